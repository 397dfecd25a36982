#!/bin/bash
# round 5, closing A/B: the round-4 tree (ab_old/, commit 0a8f862 built in place) against this tree, interleaved on one box
cd $GRAFT_REPO_ROOT
run() { (cd $1 && timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 $2 2>/dev/null | python -c "
import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('%.3f ms  loss %.6f' % (d['ms_per_step'], d['last_loss']))"); }
for rep in 1 2 3; do
  echo "[round-4 tree] $(run ab_old "")"
  echo "[this tree   ] $(run . "--shard-steps 0 --ddp-steps 0 --script-steps 0")"
done
echo "[this tree, bf16] $(run . "--shard-steps 0 --ddp-steps 0 --script-steps 0 --dtype bf16")"
echo "[round-4 tree, bf16] $(run ab_old "--dtype bf16")"
