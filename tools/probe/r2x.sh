#!/bin/bash
# round-2 GPU run X: conv epilogue with grouped, unbranched operand loads -- full suite, same-box A/B against the previous build
mkdir -p gpurun_out/r2x
timeout 500 python -m pytest tests -m gpu -q -x > gpurun_out/r2x/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2x/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 12 --warmup 4"
for rep in 1 2 3; do
  ZS3_LIB=$GRAFT_REPO_ROOT/zs3_amd/lib/variants/libzs3hip_epold.so timeout 100 $B > gpurun_out/r2x/old_$rep.json 2>> gpurun_out/r2x/err.log
  timeout 100 $B > gpurun_out/r2x/new_$rep.json 2>> gpurun_out/r2x/err.log
done
tail -3 gpurun_out/r2x/pytest.log; for f in gpurun_out/r2x/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
