#!/bin/bash
# round-2 GPU run BB: GCN-context step with the graph generator's per-image update on a second stream (and no chain cut per
# image) -- parity tests of the GCN / GMMN steps, same-box A/B
mkdir -p gpurun_out/r2bb
timeout 400 python -m pytest tests -m gpu -q -x -k "gcn or gmmn or cluster or context or distributed" > gpurun_out/r2bb/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2bb/pytest.log
G="python bench.py --no-cpu-baseline --no-roofline --workload gcn_context --steps 8 --warmup 3"
for rep in 1 2; do
  ZS3_GCN_SIDE_STREAM=0 timeout 150 $G > gpurun_out/r2bb/inline_$rep.json 2>> gpurun_out/r2bb/err.log
  timeout 150 $G > gpurun_out/r2bb/side_$rep.json 2>> gpurun_out/r2bb/err.log
done
tail -3 gpurun_out/r2bb/pytest.log; for f in gpurun_out/r2bb/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done; tail -3 gpurun_out/r2bb/err.log
