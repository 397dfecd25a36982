#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -q -x 2>&1 | tail -3
run() { (cd $1 && timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  loss %.6f' % (d['ms_per_step'], d['last_loss']))"); }
for rep in 1 2 3; do
  echo "[old tree] $(run ab_old "")"
  echo "[new tree] $(run . "--shard-steps 0 --ddp-steps 0")"
done
