#!/bin/bash
# round 5: non-temporal loads beyond bn.hip -- conv epilogue operands, pooling / resize, misc; and the 2-byte mode with / without them
cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/zs3_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -1
ZS3_LIB=$V/libzs3hip_allnt.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -1
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
run() { env $1 timeout 300 python bench.py $F --steps 20 --warmup 5 $2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
echo "[warm] $(run X=1)"
for rep in 1 2; do
  echo "[no nt     ] $(run ZS3_LIB=$V/libzs3hip_nont.so)"
  echo "[bn nt     ] $(run X=1)"
  for v in epint poolnt miscnt allnt; do echo "[bn + $v] $(run ZS3_LIB=$V/libzs3hip_$v.so)"; done
done
echo "[bf16 no nt] $(run ZS3_LIB=$V/libzs3hip_nont.so '--dtype bf16')"
echo "[bf16 bn nt] $(run X=1 '--dtype bf16')"
echo "[bf16 all  ] $(run ZS3_LIB=$V/libzs3hip_allnt.so '--dtype bf16')"
echo "[bf16 no nt] $(run ZS3_LIB=$V/libzs3hip_nont.so '--dtype bf16')"
echo "[bf16 bn nt] $(run X=1 '--dtype bf16')"
echo "[bf16 all  ] $(run ZS3_LIB=$V/libzs3hip_allnt.so '--dtype bf16')"
