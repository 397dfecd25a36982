#!/bin/bash
# round-2 GPU run G: shared-conversion wgrad kernel -- parity tests, per-layer table old vs new, step A/B
mkdir -p gpurun_out/r2g
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py tests/test_gpu_fullsize.py -m gpu -q -x -k "conv or adjoint or bf16" --durations=3 > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2g/pytest.log
ZS3_WGRAD_CV=0 timeout 120 python tools/probe/conv_bench.py 0 wgrad > gpurun_out/r2g/cbw_old.log 2>&1
ZS3_WGRAD_CV=1 timeout 120 python tools/probe/conv_bench.py 0 wgrad > gpurun_out/r2g/cbw_new.log 2>&1
B="python bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 10 --warmup 3"
for cv in 1 0 1 0; do ZS3_WGRAD_CV=$cv timeout 120 $B > gpurun_out/r2g/sup_cv${cv}_$RANDOM.json 2> gpurun_out/r2g/sup.err; done
tail -4 gpurun_out/r2g/pytest.log; paste <(cut -c1-60 gpurun_out/r2g/cbw_old.log) <(cut -c32-60 gpurun_out/r2g/cbw_new.log) | tail -30
for f in gpurun_out/r2g/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f); done
