#!/bin/bash
# round 3, GPU call i: conv_halo with the consumer's fillers spread over the MFMA gaps
mkdir -p gpurun_out/r3i; O=gpurun_out/r3i
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py -x -q -k "halo or every_tile or bn_backward or conv_kernels_bf16 or every_network" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
ZS3_SHAPES=2,16,17,18,21,22,26,27 timeout 300 python tools/probe/conv_bench.py 31,41,42 fwd 2>&1 | grep -v amdgpu.ids | tee $O/conv_fwd.txt
bash tools/probe/ab_env.sh "ZS3_HALO=1"
