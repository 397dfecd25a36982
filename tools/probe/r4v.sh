#!/bin/bash
# round 4, GPU call v: the A/B runs of r4n / r4p / r4k again on a build whose stem is correct (last_loss printed: the network is alive)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 15 --warmup 4"
run() { env $1 timeout 300 $B $2 2>/tmp/err.log | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  loss %.4f' % (d['ms_per_step'], d['last_loss']))" || tail -5 /tmp/err.log; }
for rep in 1 2; do
  echo "[bf16x3 pipe3          ] $(run A=1 "--dtype bf16x3")"
  echo "[bf16x3 pipe2          ] $(run ZS3_IGEMM_PIPE=2 "--dtype bf16x3")"
  echo "[bf16x3 racy HEAD build] $(run ZS3_LIB=$R/zs3_amd/lib/variants/libzs3hip_nobar.so "--dtype bf16x3")"
  echo "[bf16   pipe16=2 k64=1 ] $(run A=1 "--dtype bf16")"
  echo "[bf16   pipe16=3       ] $(run ZS3_IGEMM16_PIPE=3 "--dtype bf16")"
  echo "[bf16   k64=0          ] $(run ZS3_IGEMM16_K64=0 "--dtype bf16")"
done
