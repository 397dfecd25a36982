#!/bin/bash
# round 5: the stem's weight gradient (last launch of backward, alone on the main stream) sized for the whole chip; tall slab reduction
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5t; mkdir -p $O
V=$GRAFT_REPO_ROOT/zs3_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "wgrad or stem" 2>&1 | grep -E "passed|failed|error" | tail -2
ZS3_LIB=$V/libzs3hip_stem768.so timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -k "wgrad or stem or model or step" 2>&1 | grep -E "passed|failed|error" | tail -2
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
run() { env $1 timeout 300 python bench.py $F --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do
  echo "[288 ] $(run X=1)"
  for v in 512 768 1536 3072; do echo "[$v] $(run ZS3_LIB=$V/libzs3hip_stem$v.so)"; done
done
for v in 288 768 1536; do
  L=$V/libzs3hip_stem$v.so; [ $v = 288 ] && L=$GRAFT_REPO_ROOT/zs3_amd/lib/libzs3hip.so
  ZS3_LIB=$L timeout 300 rocprofv3 --kernel-trace -d $O/p$v -o t -- python bench.py $F --steps 3 --warmup 2 > $O/p$v.log 2>&1
  python - <<PY
import sqlite3
c=sqlite3.connect('$O/p$v/t_results.db')
for r in c.execute("select name, count(*), avg(end-start)/1e3 from kernels where name like '%conv_wgrad_kernel<64, 128%' or name like '%wgrad_reduce4%' group by name"):
    print('$v', r[0][:70], r[1], round(r[2],1))
PY
done
