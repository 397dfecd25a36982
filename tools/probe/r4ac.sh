#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bilinear or upsample" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_bf16_storage.py -x -q -m gpu -k "elementwise or reproducible" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "eval or config0 or default_init" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --bf16-steps 0 --steps 5 --warmup 2 --gmmn-steps 0 --no-roofline > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt -name '*.db' | head -1) 60 | grep -i "bilinear\|ce_tile\|total kernel"
