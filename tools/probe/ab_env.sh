#!/bin/bash
# same-box interleaved A/B of bench.py under environment toggles: ab_env.sh "VAR=val" "VAR2=val2" ...  ("" = baseline)
for rep in 1 2; do
  for v in "$@"; do
    ms=$(env $v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline 2>&1 | grep "^{" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "[$v] $ms"
  done
done
