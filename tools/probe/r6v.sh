#!/bin/bash
# CU-masked side streams (functional.WGRAD_MASK_CUS / FEATURE_RESERVE_CUS): GMMN step and supervised step, interleaved with the default
Q="--steps 20 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
for k in 0 2 4 6 0 8; do
  ZS3_FEATURE_RESERVE_CUS=$k timeout 300 python bench.py --workload gmmn --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>gpurun_out/r6v_gmmn_$k.err | python tools/probe/jline.py "gmmn reserve=$k" ms_per_step value
done
for k in 0 8 16 24 0 12; do
  ZS3_WGRAD_MASK_CUS=$k timeout 300 python bench.py $Q 2>gpurun_out/r6v_sup_$k.err | python tools/probe/jline.py "supervised wgrad_mask=$k" ms_per_step value
done
