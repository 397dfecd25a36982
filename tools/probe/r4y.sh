#!/bin/bash
# host-side trims (raw stream switching for the weight-gradient side streams, contiguous fast path, lean zero_grad): host time,
# the tests that exercise the side-stream bookkeeping, and the bench in both storage forms
cd tools/probe
ZS3_STORAGE=bf16 timeout 120 python host_time.py 2>&1 | tail -1
timeout 120 python host_time.py 2>&1 | tail -1
cd ../..
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "shared or unaligned or conv_bn_act_function or dropout_fused" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "supervised or accumulation" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --gmmn-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['bf16']; print('fp32', round(d['ms_per_step'],2), d['last_loss'], ' bf16', round(b['ms_per_step'],2), b['last_loss'])"
done
