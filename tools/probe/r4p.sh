#!/bin/bash
# round 4, GPU call p: batched branch-free loading epilogue (store_tile_rows): tests, then both storage forms against the previous build
R=$GRAFT_REPO_ROOT
cd $R && timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_storage.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 15 --warmup 4"
run() { env $1 timeout 300 $B $2 2>/tmp/err.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" || tail -5 /tmp/err.log; }
for rep in 1 2 3; do
  echo "[bf16x3] $(run A=1 "--dtype bf16x3")"
  echo "[bf16  ] $(run A=1 "--dtype bf16")"
done
