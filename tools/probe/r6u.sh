#!/bin/bash
# launch-shape decisions of rounds 2-5 re-checked inside the round-6 step, one at a time against the defaults, interleaved
Q="--steps 40 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
run() { env $1 timeout 300 python tools/probe/bench_flags.py $2 -- $Q $3 2>/dev/null | python tools/probe/jline.py "$4 [$1 $2]" ms_per_step last_loss; }
for rep in 1 2; do
  run A=1 ops.PW_MAXK=512 "" supervised
  run A=1 ops.HALO_BM="'auto'" "" supervised
  run A=1 ops.HALO_BM="'256'" "" supervised
  run A=1 ops.PW_MAXK=256 "" supervised
  run A=1 ops.PW_MAXK=1024 "" supervised
  run A=1 ops.PW_FORCE=51 "" supervised
  run A=1 ops.SMALL_LAUNCH_TILES=300 "" supervised
  run ZS3_WGRAD_PW_WIDE=0 ops.PW_MAXK=512 "" supervised
  run A=1 functional.DEFER_BN_APPLY=False "" supervised
done
for rep in 1 2; do
  run A=1 ops.PW_MAXK=512 "--dtype bf16" bf16
  run A=1 ops.HALO_BM="'auto'" "--dtype bf16" bf16
  run A=1 ops.HALO_BM="'192'" "--dtype bf16" bf16
  run A=1 ops.PW16=True "--dtype bf16" bf16
  run A=1 functional.DEFER_BN_APPLY=False "--dtype bf16" bf16
done
