#!/usr/bin/env python3
"""step_gaps.py DB [stream-rank]: idle time on the busiest stream of a `rocprofv3 --kernel-trace` run (rocpd database).

For the stream with the most dispatches: busy time, the sum of the gaps between consecutive kernels, a histogram of the gaps and
the (previous kernel -> next kernel) pairs that own the most idle time.  Steps are cut at `sgd_multi_kernel`."""
import collections
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]


def main():
    c = sqlite3.connect(sys.argv[1])
    streams = list(c.execute("select stream_id, count(*) from kernels group by stream_id order by 2 desc"))
    sid = streams[int(sys.argv[2]) if len(sys.argv) > 2 else 0][0]
    rows = list(c.execute("select name, start, end from kernels where stream_id=? order by start", (sid,)))
    cuts = [i for i, r in enumerate(rows) if "sgd_multi" in r[0]]
    print(f"stream {sid}: {len(rows)} dispatches, {len(cuts)} steps (streams: {streams})")
    for a, b in zip(cuts[1:-1], cuts[2:]):
        seg = rows[a + 1:b + 1]
        busy = sum(r[2] - r[1] for r in seg)
        span = seg[-1][2] - seg[0][1]
        gaps = [(seg[i + 1][1] - seg[i][2], short(seg[i][0]), short(seg[i + 1][0])) for i in range(len(seg) - 1)]
        pos = [g for g in gaps if g[0] > 0]
        print(f"step: {len(seg)} kernels, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {sum(g[0] for g in pos) / 1e6:.2f} ms "
              f"({len(pos)} gaps, median {sorted(g[0] for g in pos)[len(pos) // 2] / 1e3:.2f} us)")
    seg = rows[cuts[-2] + 1:cuts[-1] + 1]
    gaps = [(seg[i + 1][1] - seg[i][2], short(seg[i][0]), short(seg[i + 1][0])) for i in range(len(seg) - 1)]
    hist = collections.Counter()
    for g in gaps:
        us = g[0] / 1e3
        hist["<0" if us < 0 else "0-1" if us < 1 else "1-2" if us < 2 else "2-4" if us < 4 else "4-8" if us < 8 else "8-16" if us < 16 else
             "16-64" if us < 64 else ">=64"] += 1
    print("gap histogram (us), last step:", dict(hist))
    own = collections.defaultdict(lambda: [0, 0.0])
    for g in gaps:
        if g[0] > 0:
            own[(g[1], g[2])][0] += 1
            own[(g[1], g[2])][1] += g[0] / 1e3
    for k, v in sorted(own.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {v[1]:8.1f} us over {v[0]:4d} gaps ({v[1] / v[0]:5.1f} each)  {k[0]} -> {k[1]}")
    print("largest single gaps:")
    for g in sorted(gaps, key=lambda g: -g[0])[:12]:
        print(f"  {g[0] / 1e3:8.1f} us  {g[1]} -> {g[2]}")


if __name__ == "__main__":
    main()
