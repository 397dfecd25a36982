#!/bin/bash
# round-2 GPU run W: dropout fused into the BN-apply pass -- parity (layer level + full suite), same-box A/B
mkdir -p gpurun_out/r2w
timeout 500 python -m pytest tests -m gpu -q -x > gpurun_out/r2w/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2w/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 12 --warmup 4"
for i in 1 2 3; do
  ZS3_DROPOUT_FUSED=0 timeout 100 $B > gpurun_out/r2w/drop0_$i.json 2>> gpurun_out/r2w/err.log
  timeout 100 $B > gpurun_out/r2w/drop1_$i.json 2>> gpurun_out/r2w/err.log
done
G="python bench.py --no-cpu-baseline --no-roofline --workload gmmn --steps 8 --warmup 3"
ZS3_DROPOUT_FUSED=0 timeout 100 $G > gpurun_out/r2w/gmmn_drop0.json 2>> gpurun_out/r2w/err.log
timeout 100 $G > gpurun_out/r2w/gmmn_drop1.json 2>> gpurun_out/r2w/err.log
tail -3 gpurun_out/r2w/pytest.log; for f in gpurun_out/r2w/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
