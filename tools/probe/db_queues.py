#!/usr/bin/env python3
"""db_queues.py DB: which HIP stream ran on which hardware queue in a `rocprofv3 --kernel-trace` run (rocpd database):
per (stream, queue) the number of dispatches and the three most frequent kernels."""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else None
if q is None:
    print("columns:", cols)
    raise SystemExit(1)
pairs = collections.OrderedDict()
for s, qq, n in c.execute(f"select stream_id, {q}, name from kernels order by start"):
    d = pairs.setdefault((s, qq), collections.Counter())
    d[n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:28]] += 1
for (s, qq), d in pairs.items():
    tot = sum(d.values())
    if tot < 20:
        continue
    print(f"stream {s:3d} -> queue {qq:3d}: {tot:6d} dispatches  " + ", ".join(f"{k} x{v}" for k, v in d.most_common(3)))
