#!/bin/bash
# grid cap of the elementwise BatchNorm passes (csrc/bn.hip: ew_blocks, ZS3_EW_MAXBLOCKS) inside the step: 2-byte mode, fp32 storage, GMMN
Q="--steps 40 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
for k in 512 1024 2048 4096 8192; do
  ZS3_EW_MAXBLOCKS=$k timeout 300 python bench.py $Q --dtype bf16 2>/dev/null | python tools/probe/jline.py "bf16 [cap $k]" ms_per_step last_loss
done
done
for k in 512 4096; do
  ZS3_EW_MAXBLOCKS=$k timeout 300 python bench.py $Q 2>/dev/null | python tools/probe/jline.py "supervised [cap $k]" ms_per_step last_loss
done
for k in 16384 2048 1024; do
  ZS3_EW_MAXBLOCKS=$k timeout 300 python bench.py --workload gmmn --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python tools/probe/jline.py "gmmn [cap $k]" ms_per_step
done
