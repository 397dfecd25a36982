#!/bin/bash
# same-box interleaved A/B of the one-rank N > 1 path (ddp selftest) and the plain step: ab_ddp.sh ENVVAR value_a value_b [rounds]
v=$1; a=$2; b=$3; n=${4:-2}
Q="--steps 20 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
for i in $(seq $n); do
  for x in $a $b; do
    if [ "$x" = "unset" ]; then e="env -u $v"; else e="env $v=$x"; fi
    $e python bench.py $Q 2>/dev/null | python tools/probe/jline.py "plain $v=$x" ms_per_step
    $e python bench.py $Q --ddp-selftest --sync-bn 1 2>/dev/null | python tools/probe/jline.py "ddp   $v=$x" ms_per_step
  done
done
