"""Per-kernel totals of a rocprofv3 --kernel-trace CSV (calls, total ms, average us), the steps' launches only when a marker
kernel is given.  usage: csv_summary.py p_kernel_trace.csv [top_n]"""
import csv, sys, re, collections
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:70]
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    a = agg[short(r["Kernel_Name"])]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print(f"total kernel time {tot/1e6:.3f} ms over {len(rows)} dispatches\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"| {k} | {n} | {t/1e6:.3f} | {t/n/1e3:.1f} | {100*t/tot:.1f} |")
