#!/bin/bash
# ab_gmmn.sh "VAR=a" "VAR=b" ...: the GMMN step (configs[2]) under each environment
for e in "$@"; do
  env $e python bench.py --workload gmmn --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python tools/probe/jline.py "gmmn [$e]" ms_per_step
done
