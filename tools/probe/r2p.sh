#!/bin/bash
# round-2 GPU run P: full suite after the fused CE+upsample backward, label_order, branch-free wgrad staging, float4 dropout;
# same-box A/B of the CE fusion; GMMN bench + update timeline
mkdir -p gpurun_out/r2p
timeout 500 python -m pytest tests -m gpu -q -x > gpurun_out/r2p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2p/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 12 --warmup 4"
for i in 1 2 3; do
  ZS3_FUSE_CE=0 timeout 100 $B > gpurun_out/r2p/ce0_$i.json 2>> gpurun_out/r2p/err.log
  timeout 100 $B > gpurun_out/r2p/ce1_$i.json 2>> gpurun_out/r2p/err.log
done
G="python bench.py --no-cpu-baseline --no-roofline --workload gmmn --steps 8 --warmup 3"
for i in 1 2; do timeout 150 $G > gpurun_out/r2p/gmmn_$i.json 2>> gpurun_out/r2p/err.log; done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2p/kt -- $G --steps 2 --warmup 1 --gmmn-pipeline 0 > gpurun_out/r2p/kt.log 2>&1
find gpurun_out/r2p/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/probe/trace_gaps.py {} > gpurun_out/r2p/gaps.txt 2>&1
find gpurun_out/r2p/kt -name "*.csv" -size +20M -delete
tail -4 gpurun_out/r2p/pytest.log; for f in gpurun_out/r2p/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done; tail -10 gpurun_out/r2p/gaps.txt
