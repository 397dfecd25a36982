#!/bin/bash
# same-box interleaved A/B of the supervised step only (no sub-objects): ab_quick.sh ENVVAR value_a value_b [rounds]
v=$1; a=$2; b=$3; n=${4:-3}
Q="--steps 20 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
for i in $(seq $n); do
  for x in $a $b; do
    env $v=$x python bench.py $Q 2>/dev/null | python tools/probe/jline.py "$v=$x" ms_per_step
  done
done
