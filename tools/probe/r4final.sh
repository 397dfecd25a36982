#!/bin/bash
# end-of-round evidence: profiles, then the other bench entry points and the smoke test
bash tools/refresh_profiles.sh > gpurun_out/prof_refresh.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2
for w in "--dtype bf16" "--workload gmmn" "--workload gcn_context"; do
  timeout 200 python bench.py --no-cpu-baseline --bf16-steps 0 --gmmn-steps 0 --steps 10 --warmup 3 --no-roofline $w 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['ms_per_step'],2), round(d['value'],1), d.get('last_loss'))"
done
