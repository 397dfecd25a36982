#!/bin/bash
# round 5: release scope of the weight-gradient fork events (torch events / system scope / device scope): step time and main-stream gaps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -2
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
run() { timeout 300 python tools/probe/bench_flags.py functional.FORK_SCOPE=$1 -- $F --steps 20 --warmup 5 $2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do for sc in None 0 1; do echo "[fp32 scope $sc] $(run $sc)"; done; done
for sc in None 1; do echo "[bf16 scope $sc] $(run $sc '--dtype bf16')"; done
timeout 400 rocprofv3 --kernel-trace -d $O/prof -o dev -- python bench.py $F --steps 6 --warmup 3 > $O/prof.log 2>&1
python tools/probe/step_gaps.py $O/prof/dev_results.db | head -30
