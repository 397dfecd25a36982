#!/bin/bash
# round 4, GPU call q: same-box A/B of the BatchNorm element-wise kernels with / without branch-free operand loads (ZS3_LIB variants)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 15 --warmup 4"
run() { env $1 timeout 300 $B $2 2>/tmp/err.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" || tail -5 /tmp/err.log; }
for rep in 1 2 3; do
  echo "[bf16x3 new bn] $(run A=1 "--dtype bf16x3")"
  echo "[bf16x3 old bn] $(run ZS3_LIB=$R/zs3_amd/lib/variants/libzs3hip_bnold.so "--dtype bf16x3")"
  echo "[bf16   new bn] $(run A=1 "--dtype bf16")"
  echo "[bf16   old bn] $(run ZS3_LIB=$R/zs3_amd/lib/variants/libzs3hip_bnold.so "--dtype bf16")"
done
