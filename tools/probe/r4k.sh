#!/bin/bash
# round 4, GPU call k: three register stages for the bf16-stored-input kernel (tests, then same-box A/B in the 2-byte mode)
R=$GRAFT_REPO_ROOT
cd $R && timeout 600 python -m pytest tests/test_gpu_bf16_storage.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 15 --warmup 4 --dtype bf16"
run() { env $1 timeout 300 $B 2>/tmp/err.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" || tail -5 /tmp/err.log; }
for rep in 1 2; do
  for v in "ZS3_IGEMM16_PIPE=3" "ZS3_IGEMM16_PIPE=2" "ZS3_IGEMM16_PIPE=3 ZS3_READ_LOSS=0"; do
    echo "[$v] $(run "$v")"
  done
done
