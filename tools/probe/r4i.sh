#!/bin/bash
# round 4, GPU call i: 2-byte mode, same-box sweeps (persistent pointwise kernel on/off with 64-channel K steps; weight-gradient grid sizes)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --steps 15 --warmup 4 --dtype bf16"
run() { env $1 timeout 300 $B 2>/tmp/err.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" || tail -5 /tmp/err.log; }
for rep in 1 2; do
  for v in "A=1" "ZS3_PW16=0" "ZS3_WGRAD_PW_WGS=256" "ZS3_WGRAD_PW_WGS=192" "ZS3_WGRAD_STRIP_WGS=256" "ZS3_WGRAD_STRIP_WGS=128" "ZS3_WGRAD_CUS=128" "ZS3_WGRAD_CUS=64" "ZS3_WGRAD_STREAMS=1" "ZS3_WGRAD_STREAMS=3" "ZS3_HALO_BM=auto" "ZS3_HALO_BM=256"; do
    echo "[$v] $(run $v)"
  done
done
