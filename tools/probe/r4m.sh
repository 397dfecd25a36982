#!/bin/bash
# round 4, GPU call m: what bounds the 2-byte step?  (a) isolated 1x1 layers, two- vs three-deep branch-free prefetch; (b) the step without weight gradients
R=$GRAFT_REPO_ROOT
cd $R/tools/probe
for pp in 3 2; do echo "== bf16 fwd, ZS3_IGEMM16_PIPE=$pp"; ZS3_SHAPES=0,1,8,10,15 ZS3_IGEMM16_PIPE=$pp timeout 200 python conv_bench16.py 11,14 fwd 2>&1 | tail -7; done
echo "== ablation bf16"; ZS3_STORAGE=bf16 timeout 300 python wgrad_ablate.py 10 2>&1 | tail -4
