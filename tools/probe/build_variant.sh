#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]: libzs3hip with conv_igemm.hip recompiled under extra -D flags (A/B runs via ZS3_LIB)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
out=zs3_amd/lib/variants
mkdir -p $out
src=${ZS3_VARIANT_SRC:-conv_igemm}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Izs3_amd/csrc -Wno-unused-result "$@" -c zs3_amd/csrc/$src.hip -o $out/${src}_$name.o
objs=$(ls zs3_amd/lib/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $out/${src}_$name.o -o $out/libzs3hip_$name.so
echo built $out/libzs3hip_$name.so
