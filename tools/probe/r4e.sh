#!/bin/bash
# round 4, GPU call e: bf16-storage tests + step time of the 2-byte mode + kernel trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4e
mkdir -p $O
cd $R && timeout 600 python -m pytest tests/test_gpu_bf16_storage.py -q -x 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --steps 15 --warmup 4"
for dt in bf16 $EXTRA_DT; do for kk in 1 0; do export ZS3_IGEMM16_K64=$kk; echo "K64=$kk";
  ms=$(timeout 300 $B --dtype $dt 2>$O/err_$dt.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "[$dt] $ms ms"; done
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --steps 5 --warmup 2 --dtype bf16 > $O/kt.log 2>&1
db=$(find $O/kt -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $db 50 > $O/kt_bf16.md
find $O/kt -type f ! -name '*.md' -delete
head -${LINES_OUT:-40} $O/kt_bf16.md
tail -3 $O/err_bf16.log
