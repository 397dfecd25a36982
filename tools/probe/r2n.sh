#!/bin/bash
# round-2 GPU run N: tall BN-finalize kernels -- parity, same-box A/B of the supervised step
mkdir -p gpurun_out/r2n
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_dropin.py -m gpu -q -x -k "bn_ or finalize or fullsize or full_size or alias or supervised" > gpurun_out/r2n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2n/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 12 --warmup 4"
for i in 1 2 3; do
  ZS3_BN_FIN_TALL=100000000 timeout 100 $B > gpurun_out/r2n/old_$i.json 2>> gpurun_out/r2n/err.log
  timeout 100 $B > gpurun_out/r2n/new_$i.json 2>> gpurun_out/r2n/err.log
done
tail -3 gpurun_out/r2n/pytest.log; for f in gpurun_out/r2n/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
