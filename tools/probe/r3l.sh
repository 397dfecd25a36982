#!/bin/bash
# round 3, GPU call l: pointwise (1x1) weight-gradient kernel: parity, per-layer wgrad table with and without it
mkdir -p gpurun_out/r3l; O=gpurun_out/r3l
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "pointwise or unaligned or fwd_dgrad_wgrad" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
ZS3_WGRAD_PW=1 timeout 200 python tools/probe/conv_bench.py 0 wgrad > $O/wgrad_pw.txt 2>&1
ZS3_WGRAD_PW=0 timeout 200 python tools/probe/conv_bench.py 0 wgrad > $O/wgrad_old.txt 2>&1
paste -d'|' <(cut -c1-70 $O/wgrad_pw.txt) <(cut -c1-70 $O/wgrad_old.txt) | tail -34
