"""rocprofv3 rocpd database -> per-kernel table (calls, total, average) and, with --around NAME, the launch sequence around the
first few launches of kernels matching NAME on their stream (what sits between two dependent kernels).
   python tools/probe/db_kernels.py DB [--around bn_sync_pack] [--top 25]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tables if t.lower().startswith("kernels")] or [t for t in tables if "kernel" in t.lower()]
if not kt:
    print("tables:", tables)
    raise SystemExit(1)
kt = "kernels" if "kernels" in tables else kt[0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kt})")]
name = "name" if "name" in cols else [c for c in cols if "name" in c.lower()][0]
stream = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
rows = cur.execute(f"select {name}, count(*), sum(end - start), avg(end - start) from {kt} group by {name} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{len(rows)} kernels, {sum(r[1] for r in rows)} launches, {tot / 1e6:.2f} ms of kernel time")
for n, c, t, a in rows[:top]:
    print(f"{t / 1e6:9.3f} ms {c:6d} x {a / 1e3:9.1f} us  {n[:110]}")
for n, c, t, a in rows[top:]:
    if any(k in n.lower() for k in ("ccl", "onerank", "copy", "fill", "memset")):
        print(f"{t / 1e6:9.3f} ms {c:6d} x {a / 1e3:9.1f} us  {n[:110]}   <-")
if "--around" in sys.argv:
    pat = sys.argv[sys.argv.index("--around") + 1]
    sel = f"{name}, start, end" + (f", {stream}" if stream else "")
    seq = cur.execute(f"select {sel} from {kt} order by start").fetchall()
    hits = [i for i, r in enumerate(seq) if pat in r[0]]
    for i in hits[len(hits) // 2: len(hits) // 2 + 3]:
        print("---")
        t0 = seq[max(0, i - 2)][1]
        for r in seq[max(0, i - 2): i + 5]:
            print(f"  +{(r[1] - t0) / 1e3:8.1f} us .. +{(r[2] - t0) / 1e3:8.1f} us  stream {r[3] if stream else '?'}  {r[0][:90]}")
