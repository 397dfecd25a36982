#!/bin/bash
G="--workload gmmn --steps 30 --warmup 5 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
for n in 256 192 160 176 208 144; do
  timeout 300 python tools/probe/bench_pw_wgs.py $n -- $G 2>/dev/null | python tools/probe/jline.py "gmmn [pw wgs $n]" ms_per_step
done
done
for n in 256 192 160; do
  timeout 300 python tools/probe/bench_pw_wgs.py $n -- $G --dtype bf16 2>/dev/null | python tools/probe/jline.py "gmmn bf16 [pw wgs $n]" ms_per_step
done
