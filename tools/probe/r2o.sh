#!/bin/bash
# round-2 GPU run O: checkpoint -- full GPU suite, the default bench line (roofline + CPU baseline + GMMN report), the GCN-context
# and bf16 benches, kernel-trace stats of the supervised and GMMN commands
mkdir -p gpurun_out/r2o
timeout 500 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r2o/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2o/pytest.log
timeout 400 python bench.py > gpurun_out/r2o/bench_default.json 2> gpurun_out/r2o/bench_default.err
B="python bench.py --no-cpu-baseline"
timeout 120 $B --workload gcn_context --no-roofline --steps 6 --warmup 2 > gpurun_out/r2o/bench_gcn.json 2> gpurun_out/r2o/bench_gcn.err
timeout 120 $B --dtype bf16 --gmmn-steps 0 > gpurun_out/r2o/bench_bf16.json 2> gpurun_out/r2o/bench_bf16.err
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2o
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_sup -o p -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --gmmn-steps 0 > $O/kt_sup.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_gmmn -o p -- python $R/bench.py --no-cpu-baseline --workload gmmn --steps 4 --warmup 2 --no-roofline > $O/kt_gmmn.log 2>&1
for d in kt_sup kt_gmmn; do
  db=$(find $O/$d -name '*.db' | head -1)
  python $R/tools/rocprof_summary.py $db 45 > $O/$d.md
  find $O/$d -type f ! -name '*.md' -delete
done
cd $R
tail -5 gpurun_out/r2o/pytest.log; cat gpurun_out/r2o/bench_default.json; for f in gpurun_out/r2o/bench_gcn.json gpurun_out/r2o/bench_bf16.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
