#!/bin/bash
# round 5: the BatchNorm-backward statistics pass (colstats_kernel<1>): more partial rows (workgroups) and an unrolled row loop
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5x; mkdir -p $O
V=$GRAFT_REPO_ROOT/zs3_amd/lib/variants
ZS3_LIB=$V/libzs3hip_cs4x1024.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "bn or stats or colstats or dropout" 2>&1 | grep -E "passed|failed|error" | tail -2
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
run() { env $1 timeout 300 python bench.py $F --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do
  echo "[1x512 ] $(run X=1)"
  for v in 4x512 1x1024 4x1024 4x1536 8x1024; do echo "[$v] $(run ZS3_LIB=$V/libzs3hip_cs$v.so)"; done
done
