#!/bin/bash
# round 3, GPU call e: in-step kernel times with and without the strip-resident kernel (kernel trace of the supervised step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 5 --warmup 2"
for h in 1; do
  ZS3_HALO=$h timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_h$h -o p -- $B > $O/kt_h$h.log 2>&1
  csv=$(find $O/kt_h$h -name '*kernel_trace.csv' | head -1)
  python $R/tools/probe/csv_summary.py $csv 30 > $O/kt_h$h.md
  python $R/tools/probe/trace_streams.py $csv > $O/streams_h$h.txt 2>&1
  find $O/kt_h$h -type f ! -name '*.md' -delete
  grep '^{' $O/kt_h$h.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('halo=$h ms/step', d['ms_per_step'])"
  head -22 $O/kt_h$h.md; head -8 $O/streams_h$h.txt
done
