#!/bin/bash
# round-2 GPU run R: bilinear backward (unbranched candidate loads) parity + its duration; sweep of the CUs a wgrad launch is sized for
mkdir -p gpurun_out/r2r
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -q -x -k "bilinear or ce_ or supervised or full or decoder or train_forward" > gpurun_out/r2r/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2r/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 12 --warmup 4"
for rep in 1 2; do
  for cus in 64 96 128 176; do
    ZS3_WGRAD_CUS=$cus timeout 100 $B > gpurun_out/r2r/cus${cus}_$rep.json 2>> gpurun_out/r2r/err.log
  done
done
ZS3_WGRAD_CUS=96 ZS3_WGRAD_STREAMS=3 timeout 100 $B > gpurun_out/r2r/cus96s3_1.json 2>> gpurun_out/r2r/err.log
ZS3_WGRAD_CUS=256 ZS3_WGRAD_STREAMS=1 timeout 100 $B > gpurun_out/r2r/cus256s1_1.json 2>> gpurun_out/r2r/err.log
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2r
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_sup -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 4 --warmup 2 --gmmn-steps 0 > $O/kt_sup.log 2>&1
db=$(find $O/kt_sup -name '*.db' | head -1); python $R/tools/rocprof_summary.py $db 40 > $O/kt_sup.md; find $O/kt_sup -type f ! -name '*.md' -delete
cd $R
tail -3 gpurun_out/r2r/pytest.log; for f in gpurun_out/r2r/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done; grep "bilinear\|ce_tile\|dropout\|group_colsum\|finalize" gpurun_out/r2r/kt_sup.md
