#!/bin/bash
# round 3, GPU call a: first run of the strip-resident conv kernel (tile_cfg 41/42): parity, per-layer table, whole suite, bench
mkdir -p gpurun_out/r3a; O=gpurun_out/r3a
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "halo or every_tile or bn_backward" > $O/halo_tests.log 2>&1; echo "halo tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/halo_tests.log
timeout 300 python tools/probe/conv_bench.py 31,41,42,0 fwd > $O/conv_fwd.txt 2>&1; tail -32 $O/conv_fwd.txt
timeout 300 python tools/probe/conv_bench.py 31,41,42,0 dgrad > $O/conv_dgrad.txt 2>&1; tail -3 $O/conv_dgrad.txt
timeout 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt; tail -4 $O/gpu_tests.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
ZS3_HALO=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench_nohalo.json 2>> $O/bench.err; tail -c 400 $O/bench_nohalo.json
timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench_halo2.json 2>> $O/bench.err; tail -c 400 $O/bench_halo2.json
