#!/bin/bash
# f16x3 forward with the 2^6 weight scale: kernel tests, the default-init goldens, the step
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "f16x3 or refresh_planes or operand_path or pointwise_persistent or halo_kernel" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "default_init or supervised_step or eval_logits" -s 2>&1 | grep "default-init\|passed\|failed\|Error" | cut -c1-420
timeout 300 python bench.py --no-cpu-baseline --gmmn-steps 0 --bf16-steps 0 --steps 20 --warmup 5 --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', round(d['ms_per_step'],2), d['last_loss'])"
