#!/bin/bash
# round 4, GPU call d: step time of the three arithmetic / storage forms on one box + kernel trace of the 2-byte mode
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --steps 15 --warmup 4"
for dt in bf16x3 bf16f32 bf16; do
  ms=$(timeout 300 $B --dtype $dt 2>$O/err_$dt.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "[$dt] $ms ms"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --steps 5 --warmup 2 --dtype bf16 > $O/kt.log 2>&1
db=$(find $O/kt -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $db 50 > $O/kt_bf16.md
find $O/kt -type f ! -name '*.md' -delete
head -60 $O/kt_bf16.md
tail -3 $O/err_bf16.log
