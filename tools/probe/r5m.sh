#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_bf16_storage.py tests/test_gpu_bf16.py -q -x 2>&1 | tail -2
run() { (cd $1 && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 20 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  loss %.6f   bf16 %.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss'], d['bf16']['ms_per_step'], d['bf16']['last_loss']))"); }
for rep in 1 2; do
  echo "[old tree] $(run ab_old "")"
  echo "[new tree] $(run . "--shard-steps 0 --ddp-steps 0")"
done
