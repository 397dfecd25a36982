#!/bin/bash
# round-2 GPU run B: full GPU test suite, then same-box A/B of the GMMN step (pipeline x fused MLP) and of the wgrad stream pool
mkdir -p gpurun_out/r2b
timeout 420 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline"
for pipe in 0 1; do for fused in 0 1; do
  ZS3_GMMN_FUSED=$fused timeout 150 $B --workload gmmn --steps 6 --warmup 2 --gmmn-pipeline $pipe > gpurun_out/r2b/gmmn_p${pipe}_f${fused}.json 2> gpurun_out/r2b/gmmn_p${pipe}_f${fused}.err
done; done
for cfg in "1 256" "2 128" "3 96" "1 256"; do set -- $cfg
  ZS3_WGRAD_STREAMS=$1 ZS3_WGRAD_CUS=$2 timeout 120 $B --steps 10 --warmup 3 --gmmn-steps 0 > gpurun_out/r2b/sup_s$1_$RANDOM.json 2> gpurun_out/r2b/sup_s$1.err
done
tail -4 gpurun_out/r2b/pytest.log
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/r2b/*.json
