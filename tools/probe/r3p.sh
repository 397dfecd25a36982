#!/bin/bash
# round 3, GPU call p: upper bound of K32 steps in the strip kernel -- timing build with a barrier every other K step (wrong results)
export ZS3_SHAPES=2,7,16,26,27
for v in "" halob2; do
  if [ -n "$v" ]; then export ZS3_LIB=$PWD/zs3_amd/lib/variants/libzs3hip_$v.so; else unset ZS3_LIB; fi
  echo "[$v]"; timeout 60 python tools/probe/conv_bench.py 0 fwd 2>&1 | grep -v amdgpu
done
