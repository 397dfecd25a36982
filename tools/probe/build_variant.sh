#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]: libzs3hip with conv_igemm.hip recompiled under extra -D flags (A/B runs via ZS3_LIB)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
out=zs3_amd/lib/variants
mkdir -p $out
src=${ZS3_VARIANT_SRC:-conv_igemm}
# (round 6: kernel sources define their entry points as <entry>__impl -- csrc/gen/plan_rename.h -- and the library exports exactly the header's names)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Izs3_amd/csrc -Wno-unused-result -include zs3_amd/csrc/gen/plan_rename.h "$@" -c zs3_amd/csrc/$src.hip -o $out/${src}_$name.o
objs=$(ls zs3_amd/lib/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=zs3_amd/lib/obj/exports.map $objs $out/${src}_$name.o -o $out/libzs3hip_$name.so
echo built $out/libzs3hip_$name.so
