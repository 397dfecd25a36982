#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/zs3_amd/lib/variants
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "every_tile or f16x3 or in_the_consumers" 2>&1 | tail -2
run() { env $1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline --shard-steps 0 --ddp-steps 0 --bf16-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2 3; do
  echo "[paired  ] $(run X=1)"
  echo "[unpaired] $(run ZS3_LIB=$V/libzs3hip_pwnopair.so)"
done
timeout 200 python tools/probe/step_layers.py 3 2>/dev/null | grep conv_pw | head -4
