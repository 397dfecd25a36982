#!/bin/bash
# round 5: the three heavy ASPP weight gradients (2048 -> 256, 3x3) held back and released next to the layer2 / layer1 end of backward
cd "$GRAFT_REPO_ROOT"
python -c "
import zs3_amd.functional as F; F.WGRAD_HOLD = True
import pytest, sys; sys.exit(pytest.main(['tests/test_gpu_model.py', 'tests/test_gpu_world2.py', 'tests/test_gpu_distributed.py', '-q', '-x']))" 2>&1 | grep -E "passed|failed|error" | tail -2
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
run() { timeout 300 python tools/probe/bench_flags.py $1 -- $F --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do
  echo "[no hold          ] $(run functional.WGRAD_HOLD=False)"
  echo "[hold 60000 / 2   ] $(run functional.WGRAD_HOLD=True)"
  echo "[hold 60000 / 4   ] $(run 'functional.WGRAD_HOLD=True functional.WGRAD_RELEASE_EVERY=4')"
  echo "[hold 100000 / 2  ] $(run 'functional.WGRAD_HOLD=True functional.WGRAD_RELEASE_ROWS=100000')"
  echo "[hold 100000 / 1  ] $(run 'functional.WGRAD_HOLD=True functional.WGRAD_RELEASE_ROWS=100000 functional.WGRAD_RELEASE_EVERY=1')"
  echo "[hold 17000 / 8   ] $(run 'functional.WGRAD_HOLD=True functional.WGRAD_RELEASE_ROWS=17000 functional.WGRAD_RELEASE_EVERY=8')"
done
