#!/bin/bash
# round 3, GPU call h: per-layer tables for profiles/r3_conv_layers.md (isolated launches) + the strip kernels' per-wave cycle split
mkdir -p gpurun_out/r3h; O=gpurun_out/r3h
timeout 300 python tools/probe/conv_bench.py 31,41,42,0 fwd 2>&1 | grep -v amdgpu.ids > $O/fwd.txt
timeout 300 python tools/probe/conv_bench.py 31,41,0 dgrad 2>&1 | grep -v amdgpu.ids > $O/dgrad.txt
ZS3_WGRAD_STRIP=0 timeout 300 python tools/probe/conv_bench.py 0 wgrad 2>&1 | grep -v amdgpu.ids > $O/wgrad_old.txt
timeout 300 python tools/probe/conv_bench.py 0 wgrad 2>&1 | grep -v amdgpu.ids > $O/wgrad_new.txt
ZS3_LIB=zs3_amd/lib/variants/libzs3hip_timing.so timeout 120 python tools/probe/halo_timing.py 2>&1 | grep -v amdgpu.ids > $O/halo_timing.txt
tail -1 $O/fwd.txt; tail -1 $O/dgrad.txt; tail -1 $O/wgrad_old.txt; tail -1 $O/wgrad_new.txt
