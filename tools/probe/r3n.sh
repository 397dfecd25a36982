#!/bin/bash
# round 3, GPU call n: where the persistent pointwise kernel's time goes (compile-time ablations, wrong results)
export ZS3_PW=0
export ZS3_SHAPES=0,1,20
for v in pwab32 pwab64; do
  if [ -n "$v" ]; then export ZS3_LIB=$PWD/zs3_amd/lib/variants/libzs3hip_$v.so; else unset ZS3_LIB; fi
  echo "[$v] "; timeout 100 python tools/probe/conv_bench.py 51,52 fwd 2>&1 | grep "33^2"
done
