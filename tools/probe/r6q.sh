#!/bin/bash
# single conv geometries moved to another kernel inside the step (ops.TILE_OVERRIDE: (m, ncols, cin, taps, stride, dgrad, store-only, io) -> tile_cfg)
Q="--steps 40 --warmup 5 --gmmn-steps 0 --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0 --no-cpu-baseline --no-roofline"
run() { timeout 300 python tools/probe/bench_flags.py "ops.TILE_OVERRIDE=$1" -- $Q $2 2>/dev/null | python tools/probe/jline.py "$3 [$1]" ms_per_step last_loss; }
for rep in 1 2; do
  run "{}" "" supervised
  run "{(17424,256,1024,1,1,1,0,0):11}" "" supervised
  run "{(17424,256,1024,1,1,0,1,0):11}" "" supervised
  run "{(17424,512,2048,1,1,1,0,0):11}" "" supervised
  run "{(17424,2048,512,1,1,1,0,0):11}" "" supervised
  run "{(67600,128,512,1,1,1,0,0):11}" "" supervised
  run "{(17424,1024,256,1,1,1,0,0):31}" "" supervised
  run "{(266256,64,64,9,1,1,0,0):11,(266256,64,64,9,1,0,1,0):11}" "" supervised
done
