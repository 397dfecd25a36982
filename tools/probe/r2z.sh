#!/bin/bash
# round-2 final GPU run: smoke, full suite, the default bench line, the other workloads, and the raw material of profiles/
mkdir -p gpurun_out/r2z
timeout 200 python __graft_entry__.py --smoke > gpurun_out/r2z/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2z/smoke.log
timeout 500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r2z/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2z/pytest.log
timeout 400 python bench.py > gpurun_out/r2z/bench_default.json 2> gpurun_out/r2z/bench_default.err
B="python bench.py --no-cpu-baseline"
timeout 120 $B --workload gmmn --no-roofline --steps 10 --warmup 3 > gpurun_out/r2z/bench_gmmn.json 2> gpurun_out/r2z/bench_gmmn.err
timeout 120 $B --workload gcn_context --no-roofline --steps 6 --warmup 2 > gpurun_out/r2z/bench_gcn.json 2> gpurun_out/r2z/bench_gcn.err
timeout 120 $B --dtype bf16 --gmmn-steps 0 > gpurun_out/r2z/bench_bf16.json 2> gpurun_out/r2z/bench_bf16.err
timeout 120 $B --host-batches --gmmn-steps 0 --no-roofline > gpurun_out/r2z/bench_hostbatches.json 2> gpurun_out/r2z/bench_hostbatches.err
timeout 120 $B --ddp-selftest --gmmn-steps 0 --no-roofline > gpurun_out/r2z/bench_ddp1.json 2> gpurun_out/r2z/bench_ddp1.err
timeout 900 bash tools/refresh_profiles.sh > gpurun_out/r2z/refresh.log 2>&1
tail -2 gpurun_out/r2z/smoke.log; tail -3 gpurun_out/r2z/pytest.log; cat gpurun_out/r2z/bench_default.json | cut -c1-600
for f in gpurun_out/r2z/bench_*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -2 | tr '\n' ' '); done
ls gpurun_out/prof
