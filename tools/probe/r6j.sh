#!/bin/bash
# HIP runtime environment knobs on the replayed step (kernel arguments in device memory, ...), interleaved
Q="--steps 40 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
G="--workload gmmn --steps 30 --warmup 5 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
for e in "A=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0"; do
  env $e timeout 300 python bench.py $Q 2>/dev/null | python tools/probe/jline.py "supervised [$e]" ms_per_step last_loss
  env $e timeout 300 python bench.py $G 2>/dev/null | python tools/probe/jline.py "gmmn       [$e]" ms_per_step
done
done
