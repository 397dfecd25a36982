#!/bin/bash
# occupancy via __launch_bounds__(256, n) on the register-staged conv kernel: default build (64x64: 4 per CU, 128x128 A16: 2 per CU)
# against the compiler's own allocation (wpe1) and 64x64 only (wpe64only); both storage forms, same box, interleaved
V=zs3_amd/lib/variants
run() {  # $1 = label, $2 = ZS3_LIB or empty
  env ${2:+ZS3_LIB=$2} timeout 300 python bench.py --no-cpu-baseline --gmmn-steps 0 --steps 20 --warmup 5 --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['bf16']; print('$1', 'fp32', round(d['ms_per_step'],2), d['last_loss'], ' bf16', round(b['ms_per_step'],2), b['last_loss'])"
}
for i in 1 2; do
  run "default(4,2)" ""
  run "wpe1(1,1)   " $PWD/$V/libzs3hip_wpe1.so
  run "wpe64only   " $PWD/$V/libzs3hip_wpe64only.so
done
