#!/bin/bash
# round-2 GPU run D: new-kernel tests (stream-K, GMMN kernels / table mode), per-layer conv table 31 vs 32, step A/Bs, GMMN timeline
mkdir -p gpurun_out/r2d
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gmmn_kernels.py tests/test_gpu_bf16.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py -m gpu -q -x --durations=5 > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d/pytest.log
timeout 120 python tools/probe/conv_bench.py 31,32 fwd > gpurun_out/r2d/cb_fwd.log 2>&1
timeout 120 python tools/probe/conv_bench.py 31,32 dgrad > gpurun_out/r2d/cb_dgrad.log 2>&1
B="python bench.py --no-cpu-baseline --no-roofline"
for sk in 1 0 1 0; do
  ZS3_STREAMK=$sk timeout 120 $B --steps 10 --warmup 3 --gmmn-steps 0 > gpurun_out/r2d/sup_sk${sk}_$RANDOM.json 2> gpurun_out/r2d/sup.err
done
for pipe in 0 1; do
  timeout 150 $B --workload gmmn --steps 6 --warmup 2 --gmmn-pipeline $pipe > gpurun_out/r2d/gmmn_p${pipe}.json 2> gpurun_out/r2d/gmmn_p${pipe}.err
done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2d/kt -- python bench.py --workload gmmn --steps 2 --warmup 1 --gmmn-pipeline 0 --no-cpu-baseline --no-roofline > gpurun_out/r2d/kt.log 2>&1
find gpurun_out/r2d/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/probe/trace_gaps.py {} > gpurun_out/r2d/gaps.txt 2>&1
find gpurun_out/r2d/kt -name "*.csv" -size +20M -delete
tail -5 gpurun_out/r2d/pytest.log; grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/r2d/*.json; tail -14 gpurun_out/r2d/gaps.txt
