#!/bin/bash
# the GCN-context step (gcn_trainer.GCNContextStep._feature_shaping: its feature pass keeps the whole chip) and the GMMN step, final settings
timeout 900 python -m pytest tests/test_gpu_gmmn_kernels.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -2
for rep in 1 2; do
  timeout 400 python bench.py --workload gcn_context --steps 20 --warmup 5 --script-steps 0 --no-cpu-baseline --no-roofline 2>/dev/null | python tools/probe/jline.py "gcn_context" ms_per_step value
  timeout 400 python bench.py --workload gmmn --steps 30 --warmup 5 --script-steps 0 --no-cpu-baseline --no-roofline 2>/dev/null | python tools/probe/jline.py "gmmn" ms_per_step value
done
