#!/bin/bash
# round-2 GPU run T: hoisted per-channel operands in affine_act / bn_act_bwd -- parity, same-box A/B against the previous build
mkdir -p gpurun_out/r2t
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_bf16.py -m gpu -q -x > gpurun_out/r2t/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2t/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 12 --warmup 4"
for rep in 1 2 3; do
  ZS3_LIB=$GRAFT_REPO_ROOT/zs3_amd/lib/variants/libzs3hip_bnold.so timeout 100 $B > gpurun_out/r2t/old_$rep.json 2>> gpurun_out/r2t/err.log
  timeout 100 $B > gpurun_out/r2t/new_$rep.json 2>> gpurun_out/r2t/err.log
done
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2t
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_sup -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 4 --warmup 2 --gmmn-steps 0 > $O/kt_sup.log 2>&1
db=$(find $O/kt_sup -name '*.db' | head -1); python $R/tools/rocprof_summary.py $db 40 > $O/kt_sup.md; find $O/kt_sup -type f ! -name '*.md' -delete
cd $R
tail -3 gpurun_out/r2t/pytest.log; for f in gpurun_out/r2t/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done; grep "affine_act\|bn_act_bwd\|colstats" gpurun_out/r2t/kt_sup.md
