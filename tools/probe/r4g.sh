#!/bin/bash
# round 4, GPU call g: per-layer tables in bf16 storage (rules' kernel vs register-staged vs persistent pointwise)
R=$GRAFT_REPO_ROOT
cd $R && timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_storage.py -q -x -k "loading_epilogues or bf16_storage or pointwise" 2>&1 | tail -4
cd $R/tools/probe
for m in fwd dgrad dgrad_epi; do echo "== bf16 storage $m"; timeout 300 python conv_bench16.py 0,11,14,52 $m; done
echo "== fp32 storage bf16x3 dgrad_epi"; ZS3_STORAGE=fp32 timeout 300 python conv_bench16.py 0,14,31,52 dgrad_epi
