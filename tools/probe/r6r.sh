#!/bin/bash
Q="--steps 40 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
run() { env $1 timeout 300 python tools/probe/bench_flags.py $2 -- $Q $3 2>/dev/null | python tools/probe/jline.py "$4 [$1 $2]" ms_per_step last_loss; }
for rep in 1 2; do
  run A=1 ops.PW_MAXK=512 "" supervised
  run A=1 ops.DMA_RULE=False "" supervised
  run A=1 functional.ASPP_LANES_LAST=False "" supervised
  run A=1 functional.EARLY_WGRAD_FORK=True "" supervised
  run A=1 functional.LAZY_SKIP_GRAD=False "" supervised
  run A=1 functional.FUSE_BN_BWD_STATS=False "" supervised
done
