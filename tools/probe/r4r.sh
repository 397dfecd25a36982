#!/bin/bash
# round 4, GPU call r: same-box A/B: the barrier in front of the register-staged kernel's epilogue, and the box itself (three builds, interleaved)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 15 --warmup 4"
run() { env $1 timeout 300 $B $2 2>/tmp/err.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" || tail -5 /tmp/err.log; }
for rep in 1 2; do
  echo "[bf16x3 barrier   ] $(run A=1 "--dtype bf16x3")"
  echo "[bf16x3 no barrier] $(run ZS3_LIB=$R/zs3_amd/lib/variants/libzs3hip_nobar.so "--dtype bf16x3")"
  echo "[bf16x3 pipe 2    ] $(run ZS3_IGEMM_PIPE=2 "--dtype bf16x3")"
  echo "[bf16   barrier   ] $(run A=1 "--dtype bf16")"
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
