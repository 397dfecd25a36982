#!/bin/bash
# round 5: shapes of the BatchNorm finalize kernels (channels x row groups per workgroup), same-box A/B of the supervised step
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/zs3_amd/lib/variants


run() { env $1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline --shard-steps 0 --ddp-steps 0 --bf16-steps 0 --script-steps 0 2>/dev/null | python -c "
import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('%.3f ms  loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do
  echo "[32x8   ] $(run X=1)"
  for v in 16x64 16x32t2048 8x64t2048 16x64t1000000 16x64t1024; do echo "[$v] $(run ZS3_LIB=$V/libzs3hip_fin$v.so)"; done
done
