#!/bin/bash
# round 3, GPU call o: register-direct epilogue in the strip-resident kernel: parity, forward table, in-step A/B against the staged build
mkdir -p gpurun_out/r3o; O=gpurun_out/r3o
timeout 500 python -m pytest tests/test_gpu_ops.py -x -q -k "halo or pointwise or every_tile or every_network" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
export ZS3_SHAPES=2,5,7,16,26,27
timeout 100 python tools/probe/conv_bench.py 0 fwd 2>&1 | grep -v amdgpu
ZS3_LIB=$PWD/zs3_amd/lib/variants/libzs3hip_halostaged.so timeout 100 python tools/probe/conv_bench.py 0 fwd 2>&1 | grep -v amdgpu
unset ZS3_SHAPES
bash tools/probe/ab_env.sh ZS3_LIB=$PWD/zs3_amd/lib/variants/libzs3hip_halostaged.so ZS3_X=1
