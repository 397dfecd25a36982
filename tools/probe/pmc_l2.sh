# L2 / L1 / TA counters for representative conv layers (one rocprofv3 --pmc pass per counter set)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
while read -r layer; do
  i=$((i+1))
  j=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_TA_TCP_STATE_READ_sum"; do
    j=$((j+1))
    rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/l2_${i}_${j} -o p --output-format csv -- python $R/tools/probe/conv_one.py $layer > $R/gpurun_out/l2_${i}_${j}.log 2>&1
  done
done <<LAYERS
21 33 256 256 3 1
21 33 2048 256 3 12
11 33 256 1024 1 1
11 33 1024 256 1 1
LAYERS
python - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
for d in sorted(glob.glob(R + "/gpurun_out/l2_*_*")):
    if not os.path.isdir(d): continue
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "NO CSV", open(d + ".log").read()[-300:]); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        if "conv_igemm" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print(os.path.basename(d), {k: round(v[1] / v[0]) for k, v in agg.items()})
PY
