#!/bin/bash
# round-2 GPU run AA: length of the captured update chains (GMMN step): 32 (default so far), 64, 128
mkdir -p gpurun_out/r2aa
G="python bench.py --no-cpu-baseline --no-roofline --workload gmmn --steps 10 --warmup 3"
for rep in 1 2; do
  for n in 32 64 128; do
    ZS3_GMMN_CHAIN=$n timeout 150 $G > gpurun_out/r2aa/chain${n}_$rep.json 2>> gpurun_out/r2aa/err.log
  done
done
for f in gpurun_out/r2aa/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
