#!/bin/bash
# round 5, call 5: shorter f16 split (v_fma_mix) + pointer-increment producers in the persistent pointwise kernel
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
ZS3_LIB=$GRAFT_REPO_ROOT/zs3_amd/lib/variants/libzs3hip_pwtiming.so timeout 100 python tools/probe/pw_timing.py 2>&1 | grep -v amdgpu.ids | head -12
ZS3_SHAPES=0,3,4,8 timeout 100 python tools/probe/conv_bench.py 52,51,0 fwd 2>&1 | grep -v amdgpu.ids
(cd ab_old && ZS3_SHAPES=0,3,4,8 timeout 100 python tools/probe/conv_bench.py 52,0 fwd 2>&1 | grep -v amdgpu.ids)
run() { (cd $1 && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  loss %.6f' % (d['ms_per_step'], d['last_loss']))"); }
for rep in 1 2; do
  echo "[old tree] $(run ab_old "")"
  echo "[new tree] $(run . "--shard-steps 0 --ddp-steps 0")"
done
