#!/bin/bash
# round 4, GPU call o: fp32 storage after the prefetch fix: same-box A/B of routing choices
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 15 --warmup 4"
run() { env $1 timeout 300 $B 2>/tmp/err.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" || tail -5 /tmp/err.log; }
for rep in 1 2; do
  for v in "A=1" "ZS3_IGEMM_PIPE=2" "ZS3_DMA=0" "ZS3_PW=0" "ZS3_PW=0 ZS3_DMA=0" "ZS3_DEFER_BN=0" "ZS3_WGRAD_PW_WGS=256"; do echo "[$v] $(run "$v")"; done
done
