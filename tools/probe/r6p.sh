#!/bin/bash
# GMMN step knobs inside the round-6 step
G="--workload gmmn --steps 30 --warmup 5 --no-cpu-baseline --no-roofline"
run() { env $1 timeout 300 python tools/probe/bench_flags.py $2 -- $G 2>/dev/null | python tools/probe/jline.py "gmmn [$1 $2]" ms_per_step value; }
for rep in 1 2; do
  run A=1 gmmn_trainer.PREP_IN_FWD1=True
  run A=1 gmmn_trainer.PREP_IN_FWD1=False
  run ZS3_GMMN_CHAIN=8 gmmn_trainer.PREP_IN_FWD1=True
  run ZS3_GMMN_CHAIN=128 gmmn_trainer.PREP_IN_FWD1=True
  run A=1 functional.ASPP_CONCURRENT=False
  run ZS3_EW_MAXBLOCKS=16384 gmmn_trainer.PREP_IN_FWD1=True
done
