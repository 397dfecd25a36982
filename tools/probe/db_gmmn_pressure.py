#!/usr/bin/env python3
"""db_gmmn_pressure.py DB: which kernels of the prefetched feature pass slow the generator loop of the GMMN step?  For every kernel that is
not one of the update chain's (mlp_gemm / mmd_* / mlp_wgrad), the update-chain launches that START while it runs, per millisecond of
its run time, summed by kernel name over the second half of a `rocprofv3 --kernel-trace` run of `bench.py --workload gmmn`.  Alone the
chain starts ~105 launches per ms (6 per 57 us)."""
import bisect
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
lo, hi = c.execute("select min(start), max(end) from kernels").fetchone()
cut = lo + (hi - lo) // 2
rows = list(c.execute("select name, start, end, stream_id from kernels where start >= ? order by start", (cut,)))
chain = ("mlp_gemm_kernel", "mmd_tile_kernel", "mmd_bwd_kernel", "mlp_wgrad_kernel")
is_chain = lambda n: any(k in n for k in chain)
starts = [s for n, s, e, sid in rows if is_chain(n)]
chain_streams = {sid for n, s, e, sid in rows if is_chain(n)}
agg = {}
for n, s, e, sid in rows:
    if is_chain(n) or sid in chain_streams:    # (kernels on the chain's own stream are ordered with it: the pass's un-pipelined runs of the bench's breakdown)
        continue
    k = bisect.bisect_right(starts, e) - bisect.bisect_left(starts, s)
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
    a = agg.setdefault(short, [0, 0.0, 0])
    a[0] += 1
    a[1] += (e - s) / 1e6
    a[2] += k
print(f"{'kernel':66s} {'calls':>6s} {'ms':>8s} {'chain launches / ms':>20s}")
for name, (calls, ms, k) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{name:66s} {calls:6d} {ms:8.2f} {k / ms if ms else 0:20.1f}")
tot_ms = (rows[-1][2] - rows[0][1]) / 1e6
print(f"window {tot_ms:.1f} ms, {len(starts)} chain launches = {len(starts) / tot_ms:.1f} per ms overall")
