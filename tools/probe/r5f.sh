#!/bin/bash
# round 5: in-step layer table + interleaved A/B against the round-4 tree (3 pairs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
timeout 200 python tools/probe/step_layers.py 3 > $O/step_layers.md 2> $O/step_layers.err
grep "conv_pw" $O/step_layers.md | head -12
run() { (cd $1 && timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  loss %.6f' % (d['ms_per_step'], d['last_loss']))"); }
for rep in 1 2 3; do
  echo "[old tree] $(run ab_old "")"
  echo "[new tree] $(run . "--shard-steps 0 --ddp-steps 0")"
done
