#!/bin/bash
# round 4, GPU call f: the producers' loading epilogue (tests), then step times with / without it
R=$GRAFT_REPO_ROOT
cd $R && timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_storage.py -q -x -k "loading_epilogues or bf16_storage or pointwise" 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --steps 15 --warmup 4"
run() { env $1 timeout 300 $B --dtype $2 2>/tmp/err.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" || tail -5 /tmp/err.log; }
for rep in 1 2; do
  echo "[bf16   pw16=1 epi=1] $(run A=1 bf16)"
  echo "[bf16   pw16=1 epi=0] $(run ZS3_PW16_EPI=0 bf16)"
  echo "[bf16   pw16=0      ] $(run ZS3_PW16=0 bf16)"
  echo "[bf16x3 lepi=1      ] $(run A=1 bf16x3)"
  echo "[bf16x3 lepi=0      ] $(run ZS3_PW_LEPI=0 bf16x3)"
done
