#!/bin/bash
# round 5: ZS3_WGRAD_CUS (what the split-K choices of the gemm / LDS-DMA weight-gradient kernels aim at) re-swept on this round's step
cd "$GRAFT_REPO_ROOT"
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
run() { env $1 timeout 300 python bench.py $F --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do
  for v in 96 128 170 256; do echo "[cus $v] $(run ZS3_WGRAD_CUS=$v)"; done
done
