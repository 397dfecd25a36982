#!/bin/bash
# end-of-round evidence: profiles (kernel-trace stats, PMC passes), per-layer tables (isolated and in-step), smoke
cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh > gpurun_out/prof_refresh.log 2>&1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5final; mkdir -p $O
timeout 200 python tools/probe/step_layers.py 3 > $O/step_layers.md 2> $O/err.log
timeout 200 python tools/probe/step_layers.py 3 16 21 bf16 > $O/step_layers_bf16.md 2>> $O/err.log
for m in fwd dgrad wgrad; do timeout 200 python tools/probe/conv_bench.py 0 $m 2>&1 | grep -v amdgpu > $O/cb_$m.log; done
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2
tail -3 $O/cb_fwd.log
