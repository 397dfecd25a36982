#!/bin/bash
# round 5, call 2: full GPU suite after the ADVICE / range-guard / housekeeping changes, then the bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log
cat $O/pytest.log | tail -8
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gmmn-steps 5 --bf16-steps 10 > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json'))
print('fp32', d['ms_per_step'], d['last_loss'], 'bf16', d['bf16']['ms_per_step'], d['bf16']['last_loss'], 'gmmn', d['gmmn']['ms_per_step'])"
