#!/usr/bin/env python3
"""db_stream_pressure.py DB: for every kernel NOT on the busiest stream (the main chain), the main-chain launches that START while it runs,
per millisecond of its run time, by kernel name (second half of a `rocprofv3 --kernel-trace` run of the supervised bench).  The main
chain alone starts ~15 launches per ms (653 in 43 ms)."""
import bisect
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
lo, hi = c.execute("select min(start), max(end) from kernels").fetchone()
cut = lo + (hi - lo) // 2
rows = list(c.execute("select name, start, end, stream_id from kernels where start >= ? order by start", (cut,)))
count = {}
for n, s, e, sid in rows:
    count[sid] = count.get(sid, 0) + 1
main = max(count, key=count.get)
starts = [s for n, s, e, sid in rows if sid == main]
agg = {}
for n, s, e, sid in rows:
    if sid == main:
        continue
    k = bisect.bisect_right(starts, e) - bisect.bisect_left(starts, s)
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60] + f" [stream {sid}]"
    a = agg.setdefault(short, [0, 0.0, 0])
    a[0] += 1
    a[1] += (e - s) / 1e6
    a[2] += k
print(f"main stream {main}: {len(starts)} launches in {(rows[-1][2] - rows[0][1]) / 1e6:.1f} ms")
print(f"{'side-stream kernel':74s} {'calls':>6s} {'ms':>8s} {'avg us':>8s} {'main launches / ms':>20s}")
for name, (calls, ms, k) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{name:74s} {calls:6d} {ms:8.2f} {1e3 * ms / calls:8.1f} {k / ms if ms else 0:20.1f}")
