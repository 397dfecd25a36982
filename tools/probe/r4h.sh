#!/bin/bash
# round 4, GPU call h: in-step A/B of the producers' loading epilogue, both storage forms (same box, interleaved)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --steps 15 --warmup 4"
run() { env $1 timeout 300 $B --dtype $2 2>/tmp/err.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" || tail -5 /tmp/err.log; }
for rep in 1 2; do
  echo "[bf16x3 lepi=1 mink=384] $(run A=1 bf16x3)"
  echo "[bf16x3 lepi=0         ] $(run ZS3_PW_LEPI=0 bf16x3)"
  echo "[bf16x3 lepi=1 mink=1024] $(run ZS3_PW_LEPI_MINK=1024 bf16x3)"
  echo "[bf16   epi=1 mink=256 ] $(run A=1 bf16)"
  echo "[bf16   epi=1 mink=1024] $(run ZS3_PW16_EPI_MINK=1024 bf16)"
  echo "[bf16   epi=0          ] $(run ZS3_PW16_EPI=0 bf16)"
  echo "[bf16   pw16=0         ] $(run ZS3_PW16=0 bf16)"
done
