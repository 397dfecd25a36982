"""Per-kernel durations and the idle gap before each launch from a rocprofv3 --kernel-trace CSV (one stream's timeline):
prints (a) per-kernel-name totals: calls, busy us, gap-before us; (b) one typical generator update as a launch-by-launch
timeline."""
import csv, sys, collections
import re
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:60]
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
busy, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
prev_end = ks[0][0]
for s, e, n in ks:
    n = short(n)
    busy[n] += e - s; gap[n] += max(0, s - prev_end); cnt[n] += 1; prev_end = max(prev_end, e)
print(f"{'kernel':62s} {'calls':>6s} {'busy_us':>9s} {'avg_us':>7s} {'gap_us':>9s} {'avg_gap':>7s}")
for n, b in busy.most_common(40):
    print(f"{n:62s} {cnt[n]:6d} {b/1e3:9.1f} {b/1e3/cnt[n]:7.2f} {gap[n]/1e3:9.1f} {gap[n]/1e3/cnt[n]:7.2f}")
print("total busy ms", sum(busy.values())/1e6, "total gap ms", sum(gap.values())/1e6, "span ms", (ks[-1][1]-ks[0][0])/1e6)
# one update: find the last occurrence of the epilogue kernel and print the 30 launches before it
idx = [i for i, k in enumerate(ks) if "gmmn_update_epilogue" in k[2] or "mlp_wgrad_kernel<true>" in k[2]]
if len(idx) > 10:
    i1, i0 = idx[-5], idx[-6]
    print("\none generator update (launch-by-launch): start_us dur_us gap_us name")
    for j in range(i0 + 1, i1 + 1):
        s, e, n = ks[j]
        print(f"{(s-ks[i0+1][0])/1e3:8.1f} {(e-s)/1e3:7.2f} {(s-max(k[1] for k in ks[max(0,j-3):j]))/1e3:7.2f}  {short(n)}")
    print("update period us", (ks[i1][1]-ks[i0][1])/1e3)
