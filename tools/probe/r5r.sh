#!/bin/bash
# round 5: weight-gradient launches sharing a fork event (functional.WGRAD_FORK_GROUP): tests, step time per group size, main-stream gaps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_world2.py tests/test_gpu_distributed.py tests/test_gpu_dropin.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -2
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
run() { timeout 300 python tools/probe/bench_flags.py functional.WGRAD_FORK_GROUP=$1 -- $F --steps 20 --warmup 5 $2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do for g in 1 2 4 8; do echo "[fp32 group $g] $(run $g)"; done; done
for g in 1 4; do echo "[bf16 group $g] $(run $g '--dtype bf16')"; done
for g in 1 4; do echo "[ddp  group $g] $(run $g '--ddp-selftest --sync-bn 1')"; done
timeout 400 rocprofv3 --kernel-trace -d $O/prof -o g4 -- python bench.py $F --steps 6 --warmup 3 > $O/prof.log 2>&1
python tools/probe/step_gaps.py $O/prof/g4_results.db | head -24
