#!/bin/bash
# round 5: finalize shape 16x64 as the default (tests), then the plain step's kernel trace -> idle time on the main stream
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_distributed.py -q -x 2>&1 | tail -2
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
timeout 300 python bench.py $F --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])"
timeout 400 rocprofv3 --kernel-trace -d $O/prof -o plain -- python bench.py $F --steps 6 --warmup 3 > $O/prof.log 2>&1
grep '^{' $O/prof.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('traced', d['ms_per_step'])"
python tools/probe/step_gaps.py $O/prof/plain_results.db
