#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "capped_grid" 2>&1 | grep -E "passed|failed|Error" | tail -2
for rep in 1 2 3; do
  timeout 400 python bench.py --workload gmmn --steps 30 --warmup 8 --script-steps 0 --no-cpu-baseline --no-roofline 2>/dev/null | python tools/probe/jline.py "gmmn" ms_per_step value
done
timeout 400 python tools/probe/bench_flags.py gmmn_trainer.FEATURE_ADAPT=False -- --workload gmmn --steps 30 --warmup 8 --script-steps 0 --no-cpu-baseline --no-roofline 2>/dev/null | python tools/probe/jline.py "gmmn fixed caps" ms_per_step value
timeout 300 python tools/probe/gmmn_ticks.py --workload gmmn --steps 20 --warmup 8 --script-steps 0 --no-cpu-baseline --no-roofline 2>&1 >/dev/null | tail -10
