#!/bin/bash
# resident workgroups of the persistent pointwise kernel inside the step (default 256 -> 512 for the 128-row tiles)
Q="--steps 40 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
G="--workload gmmn --steps 30 --warmup 5 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
for n in 256 128 192 224; do
  timeout 300 python tools/probe/bench_pw_wgs.py $n -- $Q 2>/dev/null | python tools/probe/jline.py "supervised [pw wgs $n]" ms_per_step last_loss
  timeout 300 python tools/probe/bench_pw_wgs.py $n -- $G 2>/dev/null | python tools/probe/jline.py "gmmn       [pw wgs $n]" ms_per_step
done
done
