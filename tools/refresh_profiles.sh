#!/bin/bash
# Runs on the GPU box: regenerates the raw material for profiles/ -- kernel-trace stats of the three bench commands (supervised
# bf16x3, supervised bf16, GMMN), the two HBM-traffic PMC passes and the MFMA-busy PMC pass.  Outputs under gpurun_out/prof/;
# copy the *.md / *.json into profiles/rNN_* (tools/rocprof_summary.py, tools/pmc_traffic.py, tools/pmc_mfma.py format them).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0"
rm -rf $O/kt_* $O/pmcb_*
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_sup -o p -- $B --steps 5 --warmup 2 --gmmn-steps 0 > $O/kt_sup.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_bf16 -o p -- $B --steps 5 --warmup 2 --gmmn-steps 0 --dtype bf16 > $O/kt_bf16.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_gmmn -o p -- $B --workload gmmn --steps 4 --warmup 2 --no-roofline > $O/kt_gmmn.log 2>&1
for d in kt_sup kt_bf16 kt_gmmn; do
  db=$(find $O/$d -name '*.db' | head -1)
  python $R/tools/rocprof_summary.py $db 45 > $O/$d.md
  find $O/$d -type f ! -name '*.md' -delete
done
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/pmcb_$set -o p --output-format csv -- $B --steps 2 --warmup 1 --gmmn-steps 0 --no-roofline > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $O 0 > $O/pmc_traffic.json
# the same two passes for the 2-byte mode (round 4)
mkdir -p $O/b16
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/b16/pmcb_$set -o p --output-format csv -- $B --steps 2 --warmup 1 --gmmn-steps 0 --no-roofline --dtype bf16 > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $O/b16 0 > $O/pmc_traffic_bf16.json
find $O/b16 -type f -delete
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $O/pmcb_MFMA -o p --output-format csv -- $B --steps 2 --warmup 1 --gmmn-steps 0 --no-roofline > $O/pmc_mfma.log 2>&1
python $R/tools/pmc_mfma.py $O/pmcb_MFMA/p_counter_collection.csv 0 > $O/pmc_mfma.md
find $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE $O/pmcb_MFMA -type f -delete
grep '^{' $O/kt_sup.log | tail -1 > $O/bench_sup.json; grep '^{' $O/kt_bf16.log | tail -1 > $O/bench_bf16.json; grep '^{' $O/kt_gmmn.log | tail -1 > $O/bench_gmmn.json
tail -30 $O/pmc_mfma.md
