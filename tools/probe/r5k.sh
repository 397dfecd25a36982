#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_bf16_storage.py -q -x 2>&1 | tail -2
run() { timeout 200 python tools/probe/bench_flags.py $1 -- --steps 20 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline --shard-steps 0 --ddp-steps 0 --bf16-steps 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  loss %.6f   bf16 %.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss'], d['bf16']['ms_per_step'], d['bf16']['last_loss']))"; }
for rep in 1 2 3; do
  echo "[split dgrad on ] $(run functional.SPLIT_RAGGED_DGRAD=True)"
  echo "[split dgrad off] $(run functional.SPLIT_RAGGED_DGRAD=False)"
done
