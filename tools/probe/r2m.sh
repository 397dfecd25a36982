#!/bin/bash
# round-2 GPU run M: host-bound check of the supervised step + GMMN update timeline after the wgrad prefetch
mkdir -p gpurun_out/r2m
timeout 200 python tools/probe/host_time.py > gpurun_out/r2m/host.txt 2>&1
timeout 200 python -m pytest tests/test_gpu_gmmn_kernels.py tests/test_gpu_dropin.py -m gpu -q -x > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2m/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline --workload gmmn --steps 8 --warmup 3"
timeout 150 $B > gpurun_out/r2m/gmmn_1.json 2> gpurun_out/r2m/gmmn.err
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2m/kt -- $B --steps 2 --warmup 1 --gmmn-pipeline 0 > gpurun_out/r2m/kt.log 2>&1
find gpurun_out/r2m/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/probe/trace_gaps.py {} > gpurun_out/r2m/gaps.txt 2>&1
find gpurun_out/r2m/kt -name "*.csv" -size +20M -delete
cat gpurun_out/r2m/host.txt; tail -3 gpurun_out/r2m/pytest.log; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2m/gmmn_1.json | head -1; tail -10 gpurun_out/r2m/gaps.txt
