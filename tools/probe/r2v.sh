#!/bin/bash
# round-2 GPU run V: tile-rule variants re-checked inside the step with the wgrad streams at 96 CUs (same-box, interleaved).
# ZS3_TILE_VARIANT was a temporary switch in ops.pick_tile (A: short-K wide layers -> cfg 31, B: -> cfg 11, C / D: 128x128-vs-64x64
# threshold 400 / 3000 tiles); all within +-0.2 ms of the rules in the tree, the switch was removed again.
mkdir -p gpurun_out/r2v
B="python bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 12 --warmup 4"
for rep in 1 2; do
  for v in base A B C D; do
    ZS3_TILE_VARIANT=$v timeout 100 $B > gpurun_out/r2v/${v}_$rep.json 2>> gpurun_out/r2v/err.log
  done
done
for f in gpurun_out/r2v/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
