#!/bin/bash
# adaptive shaping of the GMMN step's prefetched feature pass (gmmn_trainer.FEATURE_ADAPT) against the fixed caps and against no caps
timeout 900 python -m pytest tests/test_gpu_gmmn_kernels.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -2
run() { timeout 400 python tools/probe/bench_flags.py $1 -- --workload gmmn --steps 30 --warmup 8 --script-steps 0 --no-cpu-baseline --no-roofline $2 2>/dev/null | python tools/probe/jline.py "gmmn $2 [$1]" ms_per_step value; }
for args in "" "--batch 8" "--classes 6" "--size 321"; do
  run "gmmn_trainer.FEATURE_ADAPT=True" "$args"
  run "gmmn_trainer.FEATURE_ADAPT=False" "$args"
  run "gmmn_trainer.FEATURE_ADAPT=False gmmn_trainer.FEATURE_PW_WGS=0 gmmn_trainer.FEATURE_HALO_WGS=0" "$args"
done
