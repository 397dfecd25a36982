#!/bin/bash
# round 5: the LDS-DMA weight-gradient kernel (one 128 KB workgroup per CU) capped at N workgroups per launch -- does leaving CUs to the main chain pay?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5u; mkdir -p $O
V=$GRAFT_REPO_ROOT/zs3_amd/lib/variants
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
run() { env $1 timeout 300 python bench.py $F --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do
  echo "[rules] $(run X=1)"
  for v in 96 128 192 256; do echo "[$v] $(run ZS3_LIB=$V/libzs3hip_dma$v.so)"; done
done
