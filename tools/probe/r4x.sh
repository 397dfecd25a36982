#!/bin/bash
# f16x3 forward (round 4): the new kernel / plane / default-init tests, then a same-box A/B of the supervised step
mkdir -p gpurun_out/r4x
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "f16x3 or refresh_planes or conv_bn_act_function or operand_path" -s 2>&1 | grep -v "^$" | tail -45 > gpurun_out/r4x/t1.log
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "default_init" -s 2>&1 | grep "default-init\|passed\|failed\|Error\|assert" > gpurun_out/r4x/t2.log
for i in 1 2; do
  for f in 1 0; do
    ZS3_FWD_F16=$f timeout 300 python bench.py --no-cpu-baseline --bf16-steps 0 --gmmn-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FWD_F16=$f', round(d['ms_per_step'],2), d['last_loss'], d['roofline']['kernel'], round(d['roofline']['achieved'],1))" >> gpurun_out/r4x/ab.log
  done
done
cat gpurun_out/r4x/t1.log | tail -30; cat gpurun_out/r4x/t2.log; cat gpurun_out/r4x/ab.log
