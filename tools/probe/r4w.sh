#!/bin/bash
# round 4, GPU call w: the batched loading epilogue (store_tile_rows) against the per-row form, on a build whose stem is correct
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 15 --warmup 4"
run() { env $1 timeout 300 $B $2 2>/tmp/err.log | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  loss %.4f' % (d['ms_per_step'], d['last_loss']))" || tail -5 /tmp/err.log; }
for rep in 1 2; do
  echo "[bf16x3 batched epilogue ] $(run ZS3_IGEMM_PIPE=2 "--dtype bf16x3")"
  echo "[bf16x3 per-row epilogue ] $(run "ZS3_IGEMM_PIPE=2 ZS3_LIB=$R/zs3_amd/lib/variants/libzs3hip_oldepi.so" "--dtype bf16x3")"
  echo "[bf16   batched epilogue ] $(run A=1 "--dtype bf16")"
  echo "[bf16   per-row epilogue ] $(run ZS3_LIB=$R/zs3_amd/lib/variants/libzs3hip_oldepi.so "--dtype bf16")"
done
