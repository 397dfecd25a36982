#!/bin/bash
# ab_env2.sh "VAR1=a VAR2=b" "VAR1=c ..." ... : the plain step and the one-rank N > 1 step under each environment, interleaved
Q="--steps 20 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
for e in "$@"; do
  env $e python bench.py $Q 2>/dev/null | python tools/probe/jline.py "plain [$e]" ms_per_step
  env $e python bench.py $Q --ddp-selftest --sync-bn 1 2>/dev/null | python tools/probe/jline.py "ddp   [$e]" ms_per_step
done
