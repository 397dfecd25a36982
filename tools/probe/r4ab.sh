#!/bin/bash
# re-sweep of the scheduling switches after this round's kernel changes, both storage forms (the 2-byte mode never had its own sweep)
run() {
  env $1 timeout 300 python bench.py --no-cpu-baseline --gmmn-steps 0 --steps 20 --warmup 5 --no-roofline --bf16-steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['bf16']; print('%-28s' % '$1', 'fp32', round(d['ms_per_step'],2), ' bf16', round(b['ms_per_step'],2), d['last_loss'], b['last_loss'])"
}
run "X=0"
run "ZS3_HALO_BM=auto"
run "ZS3_HALO_BM=192"
run "ZS3_WGRAD_CUS=64"
run "ZS3_WGRAD_CUS=128"
run "X=0"
run "ZS3_WGRAD_PW_WGS=96"
run "ZS3_WGRAD_PW_WGS=256"
run "ZS3_WGRAD_STRIP_WGS=128"
run "ZS3_WGRAD_STRIP_WGS=256"
run "ZS3_WGRAD_STREAMS=3"
run "ZS3_WGRAD_STREAMS=1"
run "X=0"
