#!/bin/bash
# round-2 GPU run U (final): full suite with the fan-out gradient sums (Fz.fork / zs3_sum_n), their same-box A/B, the default bench
# line, the other workloads, and the raw material of profiles/ for the final code
mkdir -p gpurun_out/r2u
timeout 200 python __graft_entry__.py --smoke > gpurun_out/r2u/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2u/smoke.log
timeout 500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r2u/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2u/pytest.log
B="python bench.py --no-cpu-baseline"
for i in 1 2 3; do
  ZS3_FORK_SUM=0 timeout 100 $B --no-roofline --gmmn-steps 0 --steps 12 --warmup 4 > gpurun_out/r2u/fork0_$i.json 2>> gpurun_out/r2u/err.log
  timeout 100 $B --no-roofline --gmmn-steps 0 --steps 12 --warmup 4 > gpurun_out/r2u/fork1_$i.json 2>> gpurun_out/r2u/err.log
done
timeout 400 python bench.py > gpurun_out/r2u/bench_default.json 2> gpurun_out/r2u/bench_default.err
timeout 120 $B --workload gmmn --no-roofline --steps 10 --warmup 3 > gpurun_out/r2u/bench_gmmn.json 2> gpurun_out/r2u/bench_gmmn.err
timeout 120 $B --workload gcn_context --no-roofline --steps 6 --warmup 2 > gpurun_out/r2u/bench_gcn.json 2> gpurun_out/r2u/bench_gcn.err
timeout 120 $B --dtype bf16 --gmmn-steps 0 > gpurun_out/r2u/bench_bf16.json 2> gpurun_out/r2u/bench_bf16.err
timeout 120 $B --host-batches --gmmn-steps 0 --no-roofline > gpurun_out/r2u/bench_hostbatches.json 2> gpurun_out/r2u/bench_hostbatches.err
timeout 900 bash tools/refresh_profiles.sh > gpurun_out/r2u/refresh.log 2>&1
tail -2 gpurun_out/r2u/smoke.log; tail -3 gpurun_out/r2u/pytest.log
for f in gpurun_out/r2u/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -2 | tr '\n' ' '); done
