#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 15 --warmup 4"
run() { env $1 timeout 300 $B $2 2>/tmp/err.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" || tail -5 /tmp/err.log; }
for rep in 1 2; do
  echo "[A default lib, current source ] $(run A=1 "--dtype bf16x3")"
  echo "[B variant lib, current source ] $(run ZS3_LIB=$R/zs3_amd/lib/variants/libzs3hip_cur.so "--dtype bf16x3")"
  echo "[C variant lib, HEAD source    ] $(run ZS3_LIB=$R/zs3_amd/lib/variants/libzs3hip_nobar.so "--dtype bf16x3")"
done
