#!/bin/bash
# Runs on the GPU box: regenerates the raw material for profiles/ (kernel-trace stats of the supervised and GMMN bench commands and
# the two PMC passes).  Outputs under gpurun_out/; tools/rocprof_summary.py and tools/pmc_traffic.py turn them into profiles/*.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt_sup $O/kt_gmmn $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE
rocprofv3 --kernel-trace --stats -d $O/kt_sup -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --gmmn-steps 0 > $O/kt_sup.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_gmmn -o p -- python $R/bench.py --workload gmmn --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/kt_gmmn.log 2>&1
for d in kt_sup kt_gmmn; do
  db=$(find $O/$d -name '*.db' | head -1)
  python $R/tools/rocprof_summary.py $db 45 > $O/$d.md
  find $O/$d -type f ! -name '*.md' -delete
done
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $set --kernel-trace -d $O/pmcb_$set -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --gmmn-steps 0 --no-roofline > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $O 3 > $O/pmc_traffic.json
find $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE -type f -delete
grep '^{' $O/kt_sup.log | tail -1; grep '^{' $O/kt_gmmn.log | tail -1
