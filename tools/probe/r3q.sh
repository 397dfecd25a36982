#!/bin/bash
# round 3, GPU call q: the strip-resident kernel with bf16-STORED input (tile_cfg 141 / 142): parity, then the 3x3 forward layers in
# plain-bf16 mode with fp32 storage (41 / 42) against bf16 storage (141 / 142)
mkdir -p gpurun_out/r3q; O=gpurun_out/r3q
timeout 200 python -m pytest tests/test_gpu_bf16.py -x -q -k "bf16_stored" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
export ZS3_SHAPES=2,5,7,16,17,21,26,27 ZS3_PREC=1
timeout 100 python tools/probe/conv_bench.py 41,42 fwd 2>&1 | grep -v amdgpu | tee $O/fp32_storage.txt
ZS3_A16=1 timeout 100 python tools/probe/conv_bench.py 141,142 fwd 2>&1 | grep -v amdgpu | tee $O/bf16_storage.txt
