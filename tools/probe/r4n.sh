#!/bin/bash
# round 4, GPU call n: fp32 storage (bf16x3), register-staged kernels with the branch-free three-stage prefetch: tests + same-box A/B
R=$GRAFT_REPO_ROOT
cd $R && ZS3_IGEMM_PIPE=3 timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "conv_fwd_dgrad_wgrad or every_tile or bn_backward_sums" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 15 --warmup 4"
run() { env $1 timeout 300 $B 2>/tmp/err.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" || tail -5 /tmp/err.log; }
for rep in 1 2 3; do
  for v in "ZS3_IGEMM_PIPE=3" "ZS3_IGEMM_PIPE=2"; do echo "[$v] $(run "$v")"; done
done
