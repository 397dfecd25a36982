#!/bin/bash
# round-2 GPU run J: SGD host-side change (parity), supervised bench, per-queue timeline with context around the idle gaps
mkdir -p gpurun_out/r2j
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_distributed.py -m gpu -q -x -k "sgd or supervised or one_rank or accumulation or context_60" > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2j/pytest.log
B="python bench.py --no-cpu-baseline --gmmn-steps 0 --steps 10 --warmup 3"
for i in 1 2; do timeout 120 $B > gpurun_out/r2j/sup_$i.json 2> gpurun_out/r2j/sup.err; done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2j/kt -- python bench.py --steps 4 --warmup 2 --gmmn-steps 0 --no-cpu-baseline --no-roofline > gpurun_out/r2j/kt.log 2>&1
find gpurun_out/r2j/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/probe/trace_streams.py {} > gpurun_out/r2j/streams.txt 2>&1
find gpurun_out/r2j/kt -name "*.csv" -size +20M -delete
tail -3 gpurun_out/r2j/pytest.log; head -12 gpurun_out/r2j/streams.txt | cut -c1-260; for f in gpurun_out/r2j/sup_*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done
