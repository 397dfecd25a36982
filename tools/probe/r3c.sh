#!/bin/bash
# round 3, GPU call c (same script as b after the producer/consumer restructure): strip loads three intervals ahead + 4-slot weight ring; ablations that separate the three wave roles
mkdir -p gpurun_out/r3c; O=gpurun_out/r3b
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "halo or every_tile or bn_backward" > $O/halo_tests.log 2>&1; echo "halo tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/halo_tests.log
export ZS3_SHAPES=2,7,16,17,18,21,22,26,27
for dbg in 0 1 2 3; do
  echo "== ZS3_HALO_DEBUG=$dbg" | tee -a $O/conv_fwd.txt
  ZS3_HALO_DEBUG=$dbg timeout 300 python tools/probe/conv_bench.py 31,41,42 fwd 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_fwd.txt
done
unset ZS3_SHAPES
timeout 300 python tools/probe/conv_bench.py 31,0 dgrad > $O/conv_dgrad.txt 2>&1; tail -1 $O/conv_dgrad.txt
timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench_halo.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_halo.json'));print('halo', d['ms_per_step'], d['value'])"
ZS3_HALO=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench_nohalo.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_nohalo.json'));print('nohalo', d['ms_per_step'], d['value'])"
