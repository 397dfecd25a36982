#!/bin/bash
# kernel timeline of the GMMN step (gaps between the launches of one generator update)
mkdir -p gpurun_out/r2c; cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2c/kt -- python bench.py --workload gmmn --steps 2 --warmup 1 --gmmn-pipeline 0 --no-cpu-baseline --no-roofline > gpurun_out/r2c/kt.log 2>&1
find gpurun_out/r2c/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/probe/trace_gaps.py {} > gpurun_out/r2c/gaps.txt 2>&1
find gpurun_out/r2c/kt -name "*.csv" -size +20M -delete
tail -40 gpurun_out/r2c/gaps.txt
