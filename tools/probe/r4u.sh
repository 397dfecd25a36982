#!/bin/bash
# round 4, GPU call u: per-kernel times of two builds of the library that differ by one conditional barrier (why is one 3.5 ms per step faster?)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in A C; do
  if [ $v = C ]; then export ZS3_LIB=$R/zs3_amd/lib/variants/libzs3hip_nobar.so; else unset ZS3_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt$v -o p -- python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 5 --warmup 2 > $O/kt$v.log 2>&1
  db=$(find $O/kt$v -name '*.db' | head -1)
  python $R/tools/rocprof_summary.py $db 16 > $O/kt$v.md
  find $O/kt$v -type f ! -name '*.md' -delete
  echo "== $v"; sed -n 5,26p $O/kt$v.md
done
