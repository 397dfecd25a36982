#!/bin/bash
# grid cap of the elementwise BatchNorm passes (ZS3_EW_MAXBLOCKS; default 16384 = one float4 per thread on the big tensors)
for k in 16384 4096 2048 1024; do echo "== cap $k"; ZS3_EW_MAXBLOCKS=$k timeout 200 python tools/probe/ew_bench.py 2>/dev/null; done
Q="--steps 20 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
for k in 16384 2048 4096 16384 1024 8192; do
  ZS3_EW_MAXBLOCKS=$k timeout 300 python bench.py $Q 2>/dev/null | python tools/probe/jline.py "supervised ew_cap=$k" ms_per_step value
done
