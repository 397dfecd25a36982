#!/bin/bash
# weight-gradient launch sizes, two interleaved repetitions (defaults: ZS3_WGRAD_CUS 96, ZS3_WGRAD_PW_WGS 128, ZS3_WGRAD_STRIP_WGS 192)
Q="--steps 40 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
for e in "A=1" "ZS3_WGRAD_CUS=128" "ZS3_WGRAD_CUS=64" "ZS3_WGRAD_PW_WGS=96" "ZS3_WGRAD_PW_WGS=64" "ZS3_WGRAD_STRIP_WGS=160" "ZS3_WGRAD_STRIP_WGS=128" "ZS3_WGRAD_PW_WGS=96 ZS3_WGRAD_STRIP_WGS=160"; do
  env $e timeout 300 python bench.py $Q 2>/dev/null | python tools/probe/jline.py "supervised [$e]" ms_per_step last_loss
done
done
