#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; for i in 1 2; do python -m pytest tests/test_gpu_ops.py -q -k "stem_conv_at_full or every_network_conv_shape or every_tile" 2>&1 | tail -1; done
cd tools/probe; python stem_pipe.py /tmp/w3.pt warm
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --gmmn-steps 0 --no-roofline --bf16-steps 0 --steps 15 --warmup 4"
run() { env $1 timeout 300 $B $2 2>/tmp/err.log | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" || tail -5 /tmp/err.log; }
for rep in 1 2; do
  echo "[bf16x3 relocated ] $(run A=1 "--dtype bf16x3")"
  echo "[bf16x3 no barrier] $(run ZS3_LIB=$R/zs3_amd/lib/variants/libzs3hip_nobar.so "--dtype bf16x3")"
  echo "[bf16   relocated ] $(run A=1 "--dtype bf16")"
done
