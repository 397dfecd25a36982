#!/bin/bash
# round 3, GPU call d: conv_halo with four identical register-staging producer waves
mkdir -p gpurun_out/r3d; O=gpurun_out/r3d
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "halo or every_tile or bn_backward" > $O/halo_tests.log 2>&1; echo "halo tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/halo_tests.log
ZS3_LIB=zs3_amd/lib/variants/libzs3hip_timing.so timeout 120 python tools/probe/halo_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/timing.txt
ZS3_SHAPES=2,7,16,17,18,21,22,26,27 timeout 300 python tools/probe/conv_bench.py 31,41,42,0 fwd 2>&1 | grep -v amdgpu.ids | tee $O/conv_fwd.txt
timeout 300 python tools/probe/conv_bench.py 31,0 dgrad > $O/conv_dgrad.txt 2>&1; tail -1 $O/conv_dgrad.txt
timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench_halo.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_halo.json'));print('halo', d['ms_per_step'], d['value'])"
ZS3_HALO=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench_nohalo.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_nohalo.json'));print('nohalo', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench_halo2.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_halo2.json'));print('halo', d['ms_per_step'], d['value'])"
