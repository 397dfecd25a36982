#!/bin/bash
# round-2 GPU run S: full suite after the stem-wgrad fold; wgrad CU fine sweep; PCIe-inclusive supervised step; stem wgrad duration
mkdir -p gpurun_out/r2s
timeout 500 python -m pytest tests -m gpu -q -x > gpurun_out/r2s/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2s/pytest.log
B="python bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 12 --warmup 4"
for rep in 1 2; do
  for cus in 80 96 112; do
    ZS3_WGRAD_CUS=$cus timeout 100 $B > gpurun_out/r2s/cus${cus}_$rep.json 2>> gpurun_out/r2s/err.log
  done
done
timeout 100 $B --host-batches > gpurun_out/r2s/hostbatches_1.json 2>> gpurun_out/r2s/err.log
timeout 100 $B --host-batches > gpurun_out/r2s/hostbatches_2.json 2>> gpurun_out/r2s/err.log
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2s
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_sup -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 4 --warmup 2 --gmmn-steps 0 > $O/kt_sup.log 2>&1
db=$(find $O/kt_sup -name '*.db' | head -1); python $R/tools/rocprof_summary.py $db 40 > $O/kt_sup.md; find $O/kt_sup -type f ! -name '*.md' -delete
cd $R
tail -3 gpurun_out/r2s/pytest.log; for f in gpurun_out/r2s/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1); done; grep "conv_wgrad_kernel\|wgrad_reduce" gpurun_out/r2s/kt_sup.md
