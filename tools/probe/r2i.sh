#!/bin/bash
mkdir -p gpurun_out/r2i
B="python bench.py --no-cpu-baseline --no-roofline --workload gmmn --steps 8 --warmup 3"
for p in 1 0 1 0; do ZS3_GMMN_PRIO=$p timeout 150 $B > gpurun_out/r2i/gmmn_prio${p}_$RANDOM.json 2> gpurun_out/r2i/gmmn.err; done
timeout 200 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py tests/test_gpu_model.py -m gpu -q -x -k "gmmn or gcn" > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2i/pytest.log
tail -3 gpurun_out/r2i/pytest.log
for f in gpurun_out/r2i/gmmn_prio*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f); done
