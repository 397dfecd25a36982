#!/bin/bash
# round-2 GPU run Q: tiled CE+upsample backward (parity + same-box A/B), unbranched MMD / dgrad loads (parity + update timeline)
mkdir -p gpurun_out/r2q
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gmmn_kernels.py tests/test_gpu_dropin.py tests/test_gpu_model.py -m gpu -q -x -k "ce or mmd or gmmn or mlp or supervised or cross or alias or step" > gpurun_out/r2q/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2q/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 12 --warmup 4"
for i in 1 2 3; do
  ZS3_FUSE_CE=0 timeout 100 $B > gpurun_out/r2q/ce0_$i.json 2>> gpurun_out/r2q/err.log
  timeout 100 $B > gpurun_out/r2q/ce1_$i.json 2>> gpurun_out/r2q/err.log
done
G="python bench.py --no-cpu-baseline --no-roofline --workload gmmn --steps 8 --warmup 3"
for i in 1 2; do timeout 150 $G > gpurun_out/r2q/gmmn_$i.json 2>> gpurun_out/r2q/err.log; done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2q/kt -- $G --steps 2 --warmup 1 --gmmn-pipeline 0 > gpurun_out/r2q/kt.log 2>&1
find gpurun_out/r2q/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/probe/trace_gaps.py {} > gpurun_out/r2q/gaps.txt 2>&1
grep -h "ce_bilinear\|ce_tile\|bilinear_bwd\|label_order" gpurun_out/r2q/gaps.txt | head
find gpurun_out/r2q/kt -name "*.csv" -size +20M -delete
tail -4 gpurun_out/r2q/pytest.log; for f in gpurun_out/r2q/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done; tail -10 gpurun_out/r2q/gaps.txt
