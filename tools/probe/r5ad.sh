#!/bin/bash
# round 5: cached (plain) loads in the BatchNorm-backward statistics pass, whose inputs bn_act_bwd re-reads right behind it
cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/zs3_amd/lib/variants
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
run() { env $1 timeout 300 python bench.py $F --steps 20 --warmup 5 $2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
echo "[warm] $(run X=1)"
for rep in 1 2 3; do
  echo "[default ] $(run X=1)"
  echo "[cs cached] $(run ZS3_LIB=$V/libzs3hip_cscached.so)"
done
