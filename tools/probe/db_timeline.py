#!/usr/bin/env python3
"""db_timeline.py DB [t0_ms] [span_ms] [gap_us]: busy intervals per stream in a window of a `rocprofv3 --kernel-trace` run
(kernels closer than gap_us are merged): which streams really run side by side."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
t0_ms = float(sys.argv[2]) if len(sys.argv) > 2 else None
span = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
gap = float(sys.argv[4]) * 1e3 if len(sys.argv) > 4 else 100e3
lo, hi = c.execute("select min(start), max(end) from kernels").fetchone()
start = hi - int((span + 20) * 1e6) if t0_ms is None else lo + int(t0_ms * 1e6)
end = start + int(span * 1e6)
rows = list(c.execute("select stream_id, queue_id, name, start, end from kernels where end >= ? and start <= ? order by start", (start, end)))
by = {}
for s, q, n, a, b in rows:
    by.setdefault((s, q), []).append((a, b, n))
for (s, q), ks in sorted(by.items()):
    iv = []
    for a, b, n in ks:
        if iv and a - iv[-1][1] < gap:
            iv[-1][1] = max(iv[-1][1], b)
            iv[-1][2] += 1
            iv[-1][3] += b - a
        else:
            iv.append([a, b, 1, b - a, n])
    print(f"stream {s} (queue {q}): {len(ks)} kernels")
    for a, b, k, busy, n in iv:
        if b - a < 200e3 and k < 5:
            continue
        print(f"   {(a - start) / 1e6:8.2f} .. {(b - start) / 1e6:8.2f} ms  ({(b - a) / 1e6:6.2f} ms, {k:5d} kernels, {busy / 1e6:6.2f} ms busy)  first: "
              + n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40])
