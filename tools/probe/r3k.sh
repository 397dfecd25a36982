#!/bin/bash
# round 3, GPU call k: where the GCN-context step's time goes (kernel trace of the context flow)
mkdir -p gpurun_out/r3k; O=gpurun_out/r3k
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o gcn -- python $R/bench.py --no-cpu-baseline --no-roofline --workload gcn_context --steps 5 --warmup 2 > $R/$O/bench.json 2> $R/$O/bench.err
cd $R
f=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python tools/probe/csv_summary.py $f 25 > $O/kernel_stats.md 2>&1 || head -30 $f
cat $O/kernel_stats.md | head -40
tail -c 600 $O/bench.json
