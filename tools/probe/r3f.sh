#!/bin/bash
# round 3, GPU call f: strip-resident weight-gradient kernel: parity of every 3x3 shape, per-layer table against the old kernels
mkdir -p gpurun_out/r3f; O=gpurun_out/r3f
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "fwd_dgrad_wgrad or every_network_conv_shape or conv_bn_act_function" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
ZS3_WGRAD_STRIP=0 timeout 300 python tools/probe/conv_bench.py 0 wgrad 2>&1 | grep -v amdgpu.ids > $O/wgrad_old.txt
timeout 300 python tools/probe/conv_bench.py 0 wgrad 2>&1 | grep -v amdgpu.ids > $O/wgrad_strip.txt
paste -d'\n' $O/wgrad_old.txt $O/wgrad_strip.txt | grep -E "k3 s1|TOTAL"
