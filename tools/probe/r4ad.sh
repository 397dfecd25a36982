#!/bin/bash
# six launches per generator update (prep merged into the first GEMM): the GMMN tests, then the step with and without
timeout 600 python -m pytest tests/test_gpu_gmmn_kernels.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_world2.py tests/test_gpu_distributed.py -x -q -m gpu -k "gmmn or gcn or GMMN" 2>&1 | tail -2
for i in 1 2; do
  for f in 1 0; do
    ZS3_GMMN_PREP_FUSED=$f timeout 300 python bench.py --no-cpu-baseline --bf16-steps 0 --workload gmmn --steps 10 --warmup 3 --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PREP_FUSED=$f', round(d['ms_per_step'],2), round(d['value'],1), d.get('last_loss'))"
  done
done
