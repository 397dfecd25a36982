#!/bin/bash
# round 5: the 2-byte mode with the two-workgroup pointwise kernel on its 1x1 layers (ops.PW16), interleaved
cd $GRAFT_REPO_ROOT
run() { timeout 200 python tools/probe/bench_flags.py $1 -- --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline --shard-steps 0 --ddp-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do
  echo "[PW16 off] $(run ops.HALO16=True)"
  echo "[PW16 on ] $(run ops.PW16=True)"
done
timeout 300 python -m pytest tests/test_gpu_bf16_storage.py -q -x 2>&1 | tail -2
