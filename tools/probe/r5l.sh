#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 200 python tools/probe/bench_flags.py $1 -- --steps 3 --warmup 2 --no-cpu-baseline --gmmn-steps 0 --no-roofline --shard-steps 20 --ddp-steps 0 --bf16-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['shard']; print('shard %.3f ms  loss %.6f' % (s['ms_per_step'], s['last_loss']))"; }
for rep in 1 2 3; do
  echo "[small-launch rule on ] $(run ops.SMALL_LAUNCH_TILES=100)"
  echo "[small-launch rule off] $(run ops.SMALL_LAUNCH_TILES=0)"
done
