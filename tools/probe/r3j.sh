#!/bin/bash
# round 3, GPU call j: which launches make up the small-tile conv kernels' in-step time (kernel trace, grouped by grid size)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --gmmn-steps 0 --steps 5 --warmup 2 > $O/kt.log 2>&1
csv=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python - "$csv" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0]
    if not n.startswith("conv_igemm_kernel"): continue
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256))))
    a = agg[(n, g)]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print("kernel, workgroups, launches per step, avg us, ms per step  (7 steps traced)")
for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:38s} {g:6d} {c/7:6.1f} {t/c/1e3:8.1f} {t/7/1e6:7.3f}")
print("total ms per step", tot / 7 / 1e6)
PY
find $O/kt -type f -delete
