#!/bin/bash
# round 4, GPU call l: per-queue busy / idle of one 2-byte-mode step (is the main chain waiting for the host?)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 6 --warmup 3 --dtype ${DT:-bf16} > $O/kt.log 2>&1
csv=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python $R/tools/probe/trace_streams.py $csv 2>&1 | head -60
find $O/kt -type f -delete
