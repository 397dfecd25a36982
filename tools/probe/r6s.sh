#!/bin/bash
# kernel-family switches re-checked inside the round-6 step (environment switches of DESIGN.md section 9), interleaved with the default
Q="--steps 40 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
for e in "A=1" "ZS3_WGRAD_PW=0" "ZS3_WGRAD_STRIP=0" "ZS3_PW=0" "ZS3_WGRAD_PW_WGS=192" "ZS3_WGRAD_STRIP_WGS=256" "ZS3_WGRAD_CUS=192"; do
  env $e timeout 300 python bench.py $Q 2>/dev/null | python tools/probe/jline.py "supervised [$e]" ms_per_step last_loss
done
done
for e in "A=1" "ZS3_WGRAD_PW=0" "ZS3_WGRAD_STRIP=0" "ZS3_WGRAD_PW_WGS=192" "A=1" "ZS3_WGRAD_CUS=192" "ZS3_WGRAD_STRIP_WGS=256"; do
  env $e timeout 300 python bench.py $Q --dtype bf16 2>/dev/null | python tools/probe/jline.py "bf16 [$e]" ms_per_step last_loss
done
