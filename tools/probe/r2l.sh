#!/bin/bash
# round-2 GPU run L: float4 Adam epilogue, burst wgrad loads, parallel loss sum -- parity tests, bench, update timeline
mkdir -p gpurun_out/r2l
timeout 300 python -m pytest tests/test_gpu_gmmn_kernels.py tests/test_gpu_model.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py tests/test_gpu_ops.py -m gpu -q -x -k "gmmn or gcn or mmd or mlp or optimizer_state" --durations=4 > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline --workload gmmn --steps 8 --warmup 3"
for i in 1 2; do timeout 150 $B > gpurun_out/r2l/gmmn_$i.json 2> gpurun_out/r2l/gmmn.err; done
timeout 150 $B --gmmn-pipeline 0 > gpurun_out/r2l/gmmn_nopipe.json 2>> gpurun_out/r2l/gmmn.err
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2l/kt -- $B --steps 2 --warmup 1 --gmmn-pipeline 0 > gpurun_out/r2l/kt.log 2>&1
find gpurun_out/r2l/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/probe/trace_gaps.py {} > gpurun_out/r2l/gaps.txt 2>&1
find gpurun_out/r2l/kt -name "*.csv" -size +20M -delete
tail -4 gpurun_out/r2l/pytest.log; for f in gpurun_out/r2l/gmmn*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done; tail -12 gpurun_out/r2l/gaps.txt
