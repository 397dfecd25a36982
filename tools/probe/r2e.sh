#!/bin/bash
# round-2 GPU run E: full GPU suite, both bench lines (no CPU baseline), per-queue timeline of the supervised step
mkdir -p gpurun_out/r2e
timeout 420 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2e/pytest.log
B="python bench.py --no-cpu-baseline"
timeout 200 $B --steps 10 --warmup 3 > gpurun_out/r2e/bench_x3.json 2> gpurun_out/r2e/bench_x3.err
timeout 200 $B --steps 10 --warmup 3 --dtype bf16 --gmmn-steps 0 > gpurun_out/r2e/bench_bf16.json 2> gpurun_out/r2e/bench_bf16.err
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2e/kt -- python bench.py --steps 4 --warmup 2 --gmmn-steps 0 --no-cpu-baseline --no-roofline > gpurun_out/r2e/kt.log 2>&1
find gpurun_out/r2e/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/probe/trace_streams.py {} > gpurun_out/r2e/streams.txt 2>&1
find gpurun_out/r2e/kt -name "*.csv" -size +20M -delete
tail -6 gpurun_out/r2e/pytest.log; cat gpurun_out/r2e/streams.txt; grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/r2e/*.json
