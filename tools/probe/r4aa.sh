#!/bin/bash
# after the barrier in front of the BN-sum epilogue: is the step bit-reproducible in both storage forms, and what does it cost?
cd tools/probe
ZS3_STORAGE=bf16 timeout 120 python determinism.py 4 2>&1 | grep repeat
timeout 120 python determinism.py 3 2>&1 | grep repeat
ZS3_STORAGE=bf16 timeout 200 python nanfill.py 2>&1 | tail -4 | cut -c1-300
cd ../..
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stem or every_tile or f16x3" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --gmmn-steps 0 --steps 20 --warmup 5 --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['bf16']; print('fp32', round(d['ms_per_step'],2), d['last_loss'], ' bf16', round(b['ms_per_step'],2), b['last_loss'])"
done
