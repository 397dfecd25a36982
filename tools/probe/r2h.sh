#!/bin/bash
# round-2 GPU run H: CU-masked feature stream for the pipelined GMMN step (reserved CUs 0 / 16 / 32 / 64), MMD change parity
mkdir -p gpurun_out/r2h
timeout 200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gmmn_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "mmd or gmmn or gcn" --durations=3 > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2h/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline --workload gmmn --steps 8 --warmup 3"
for r in 0 32 16 64 0 32; do
  ZS3_GMMN_RESERVED_CUS=$r timeout 150 $B > gpurun_out/r2h/gmmn_r${r}_$RANDOM.json 2> gpurun_out/r2h/gmmn.err
done
timeout 150 $B --gmmn-pipeline 0 > gpurun_out/r2h/gmmn_nopipe.json 2>> gpurun_out/r2h/gmmn.err
tail -3 gpurun_out/r2h/pytest.log
for f in gpurun_out/r2h/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f); done
