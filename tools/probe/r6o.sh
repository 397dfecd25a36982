#!/bin/bash
G="--workload gmmn --steps 30 --warmup 5 --no-cpu-baseline --no-roofline"
run() { env $1 timeout 300 python tools/probe/bench_flags.py $2 -- $G $3 2>/dev/null | python tools/probe/jline.py "gmmn [$1 $2 $3]" ms_per_step value; }
for rep in 1 2; do
  run A=1 gmmn_trainer.FEATURE_LANES=False ""
  run A=1 gmmn_trainer.FEATURE_LANES=True ""
  run A=1 gmmn_trainer.FEATURE_LANES=False "--gmmn-pipeline 0"
  run A=1 gmmn_trainer.FEATURE_LANES=True "--gmmn-pipeline 0"
  run ZS3_GMMN_CHAIN=8 gmmn_trainer.FEATURE_LANES=False ""
  run A=1 gmmn_trainer.FEATURE_LANES=False "--dtype bf16"
  run A=1 gmmn_trainer.FEATURE_LANES=True "--dtype bf16"
done
ZS3_GMMN_TICKS=1 timeout 200 python tools/probe/gmmn_ticks.py 2>/dev/null | tail -12
