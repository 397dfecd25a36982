#!/bin/bash
# round-2 GPU run EE: packed bf16 activation planes as the A operand of the LDS-DMA conv (tile_cfg 33): parity, per-layer forward
# times against tile_cfg 31, and a same-box check that the default kernel (same source, new template parameter) did not move
mkdir -p gpurun_out/r2ee
timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "planes or every_tile or fwd_dgrad" > gpurun_out/r2ee/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2ee/pytest.log
timeout 150 python tools/probe/conv_bench.py 31,33 fwd > gpurun_out/r2ee/conv_bench_fwd.txt 2>&1
ZS3_PREC=1 timeout 150 python tools/probe/conv_bench.py 31,33 fwd > gpurun_out/r2ee/conv_bench_fwd_bf16.txt 2>&1
B="python bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 12 --warmup 4"
for rep in 1 2; do
  ZS3_LIB=$GRAFT_REPO_ROOT/zs3_amd/lib/variants/libzs3hip_old.so timeout 100 $B > gpurun_out/r2ee/old_$rep.json 2>> gpurun_out/r2ee/err.log
  timeout 100 $B > gpurun_out/r2ee/new_$rep.json 2>> gpurun_out/r2ee/err.log
done
grep -E "passed|failed|rc=" gpurun_out/r2ee/pytest.log; tail -4 gpurun_out/r2ee/conv_bench_fwd.txt; tail -2 gpurun_out/r2ee/conv_bench_fwd_bf16.txt
for f in gpurun_out/r2ee/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
