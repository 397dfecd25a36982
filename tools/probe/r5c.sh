#!/bin/bash
# round 5, call 3: same-box A/B of the round-4 tree (ab_old/) against this tree; tap skipping on the LDS-DMA kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
run() { (cd $1 && env $2 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 $3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  loss %.6f' % (d['ms_per_step'], d['last_loss']))"); }
NEW="--shard-steps 0 --ddp-steps 0"
for rep in 1 2; do
  echo "[old tree       ] $(run ab_old X=0 "")"
  echo "[new tree       ] $(run . X=0 "$NEW")"
  echo "[new, FWD_F16=0 ] $(run . ZS3_FWD_F16=0 "$NEW")"
done
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "range_guard or every_tile or f16x3 or conv_fwd_dgrad" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -x -k "batch_16" 2>&1 | tail -3
ZS3_SHAPES=21,22,23 timeout 100 python tools/probe/conv_bench.py 31,41,42,0 fwd 2>&1 | tail -5
ZS3_SHAPES=21,22,23 timeout 100 python tools/probe/conv_bench.py 31,41,0 dgrad 2>&1 | tail -5
(cd ab_old && ZS3_SHAPES=21,22,23 timeout 100 python tools/probe/conv_bench.py 31 fwd 2>&1 | tail -5)
