#!/bin/bash
# round-2 GPU run CC: GCN-context step, GMMN update chain cut after every image (as before) or not; graph updates in line
mkdir -p gpurun_out/r2cc
G="python bench.py --no-cpu-baseline --no-roofline --workload gcn_context --steps 8 --warmup 3"
for rep in 1 2; do
  ZS3_GCN_FLUSH=1 timeout 150 $G > gpurun_out/r2cc/flush_$rep.json 2>> gpurun_out/r2cc/err.log
  ZS3_GCN_FLUSH=0 timeout 150 $G > gpurun_out/r2cc/noflush_$rep.json 2>> gpurun_out/r2cc/err.log
done
for f in gpurun_out/r2cc/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
