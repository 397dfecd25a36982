#!/usr/bin/env python3
"""db_window.py DB stream_id: around one of the longest idle gaps (< 20 ms) of that stream in the last 40 % of the run, the dispatches of every
stream that start within the gap's edges +- 0.3 ms (what a stalled stream was waiting behind)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
sid = int(sys.argv[2])
lo, hi = c.execute("select min(start), max(end) from kernels").fetchone()
cut = lo + int(0.6 * (hi - lo))
rows = list(c.execute("select start, end, name from kernels where stream_id=? and start>=? order by start", (sid, cut)))
gaps = sorted(((rows[i + 1][0] - rows[i][1], i) for i in range(len(rows) - 1)), reverse=True)
gaps = [x for x in gaps if x[0] < 20e6]      # (not the pauses between the phases of a bench run)
g, i = gaps[len(gaps[:6]) // 2]
a, b = rows[i][1], rows[i + 1][0]
print(f"stream {sid}: longest gap {g / 1e6:.2f} ms, from +0 to +{(b - a) / 1e6:.2f} ms")
for s, q, n, st, en in c.execute("select stream_id, queue_id, name, start, end from kernels where end>=? and start<=? order by start",
                                 (a - 300000, b + 300000)):
    near = min(abs(st - a), abs(st - b), abs(en - a), abs(en - b)) < 300000
    if near or s != 3:
        print(f"  +{(st - a) / 1e6:8.3f} .. +{(en - a) / 1e6:8.3f}  stream {s} q{q}  " + n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60])
