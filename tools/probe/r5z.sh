#!/bin/bash
# round 5: tall BatchNorm partial buffers (stem, layer1) folded on a (channel group, row slice) grid before the finalize kernels read them
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_distributed.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -2
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
run() { timeout 300 python tools/probe/bench_flags.py $1 -- $F --steps 20 --warmup 5 $2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
echo "[warm] $(run ops.BN_FOLD=False)"
for rep in 1 2 3; do
  echo "[one stage] $(run ops.BN_FOLD=False)"
  echo "[folded   ] $(run ops.BN_FOLD=True)"
done
echo "[ddp one stage] $(run ops.BN_FOLD=False '--ddp-selftest --sync-bn 1')"
echo "[ddp folded   ] $(run ops.BN_FOLD=True '--ddp-selftest --sync-bn 1')"
