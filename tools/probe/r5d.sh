#!/bin/bash
# round 5, call 4: where the persistent pointwise kernel's time goes -- ablation builds (timing probes, wrong results) of conv_pw.hip
# on the three forward shapes it serves most: 1 = no epilogue work, 2 = no MFMAs, 4 = no global loads, 8 = no split / LDS writes
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/zs3_amd/lib/variants
for v in base pwa1 pwa2 pwa4 pwa8 pwa12 pwa6 pwa14 pwa15; do
  lib=$V/libzs3hip_$v.so; [ $v = base ] && lib=$GRAFT_REPO_ROOT/zs3_amd/lib/libzs3hip.so
  echo "== $v"; ZS3_LIB=$lib ZS3_SHAPES=0,3,8 timeout 100 python tools/probe/conv_bench.py 52,51 fwd 2>&1 | grep -v amdgpu.ids | head -3
done
