#!/bin/bash
# round 5, call 1: this box's baseline -- bench line, in-step per-layer conv table, isolated per-layer tables
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gmmn-steps 5 --bf16-steps 10 > $O/bench.json 2> $O/bench.err
timeout 200 python tools/probe/step_layers.py 3 > $O/step_layers.md 2> $O/step_layers.err
timeout 200 python tools/probe/step_layers.py 3 16 21 bf16 > $O/step_layers_bf16.md 2>> $O/step_layers.err
timeout 200 python tools/probe/conv_bench.py 0 fwd > $O/cb_fwd.log 2>&1
timeout 200 python tools/probe/conv_bench.py 0 dgrad > $O/cb_dgrad.log 2>&1
tail -3 $O/bench.err; head -c 600 $O/bench.json; echo; head -30 $O/step_layers.md
