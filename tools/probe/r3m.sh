#!/bin/bash
# round 3, GPU call m: persistent pointwise conv kernel (tile_cfg 51/52): parity, then per-layer forward table against the rules
mkdir -p gpurun_out/r3m; O=gpurun_out/r3m
timeout 240 python -m pytest tests/test_gpu_ops.py -x -q -k "pointwise_persistent" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/tests.log
export ZS3_SHAPES=0,1,3,4,6,8,9,10,15,19,20,24
ZS3_PW=0 timeout 200 python tools/probe/conv_bench.py 0,51,52 fwd > $O/fwd.txt 2>&1; grep -v amdgpu $O/fwd.txt
