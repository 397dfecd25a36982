#!/bin/bash
# tile rules of the register-staged kernel re-checked inside the step (ops.TILE16_SWITCH, SHORTK_TILE, TILE_SWITCH), interleaved
Q="--steps 40 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
for f in "ops.TILE16_SWITCH=512" "ops.TILE16_SWITCH=0" "ops.TILE16_SWITCH=1200" "ops.TILE16_SWITCH=3000" "ops.TILE16_SWITCH=1000000"; do
  timeout 300 python tools/probe/bench_flags.py $f -- $Q --dtype bf16 2>/dev/null | python tools/probe/jline.py "bf16 [$f]" ms_per_step last_loss
done
done
for rep in 1 2; do
for f in "ops.SHORTK_TILE=14" "ops.SHORTK_TILE=11" "ops.TILE_SWITCH=400" "ops.TILE_SWITCH=3000"; do
  timeout 300 python tools/probe/bench_flags.py $f -- $Q 2>/dev/null | python tools/probe/jline.py "supervised [$f]" ms_per_step last_loss
done
done
