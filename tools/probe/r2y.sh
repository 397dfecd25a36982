#!/bin/bash
# round-2 GPU run Y: final measurements of the code at HEAD (the full suite passed on it in tools/probe/r2w.sh / r2x.sh): smoke, the
# default bench line, the other workloads, three plain supervised runs, and the raw material of profiles/
mkdir -p gpurun_out/r2y
timeout 200 python __graft_entry__.py --smoke > gpurun_out/r2y/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2y/smoke.log
B="python bench.py --no-cpu-baseline"
for i in 1 2 3; do timeout 100 $B --no-roofline --gmmn-steps 0 --steps 12 --warmup 4 > gpurun_out/r2y/fork1_$i.json 2>> gpurun_out/r2y/err.log; done
timeout 400 python bench.py > gpurun_out/r2y/bench_default.json 2> gpurun_out/r2y/bench_default.err
timeout 120 $B --workload gmmn --no-roofline --steps 10 --warmup 3 > gpurun_out/r2y/bench_gmmn.json 2> gpurun_out/r2y/bench_gmmn.err
timeout 120 $B --workload gcn_context --no-roofline --steps 6 --warmup 2 > gpurun_out/r2y/bench_gcn.json 2> gpurun_out/r2y/bench_gcn.err
timeout 120 $B --dtype bf16 --gmmn-steps 0 > gpurun_out/r2y/bench_bf16.json 2> gpurun_out/r2y/bench_bf16.err
timeout 120 $B --host-batches --gmmn-steps 0 --no-roofline > gpurun_out/r2y/bench_hostbatches.json 2> gpurun_out/r2y/bench_hostbatches.err
timeout 900 bash tools/refresh_profiles.sh > gpurun_out/r2y/refresh.log 2>&1
tail -2 gpurun_out/r2y/smoke.log
for f in gpurun_out/r2y/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -2 | tr '\n' ' '); done
