#!/usr/bin/env python3
"""db_diff.py DB_A DB_B [steps]: per-kernel difference of two `rocprofv3 --kernel-trace` runs of the same number of steps
(launches and kernel time per step, B minus A, largest first) -- what the N > 1 code path adds to the plain step."""
import collections
import sqlite3
import sys


def table(path):
    c = sqlite3.connect(path)
    out = collections.OrderedDict()
    for n, k, t in c.execute("select name, count(*), sum(end - start) from kernels group by name"):
        n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        a = out.setdefault(n, [0, 0])
        a[0] += k
        a[1] += t
    return out


a, b = table(sys.argv[1]), table(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = []
for n in set(a) | set(b):
    ka, ta = a.get(n, (0, 0))
    kb, tb = b.get(n, (0, 0))
    rows.append(((tb - ta) / 1e6 / steps, (kb - ka) / steps, ta / 1e6 / steps, tb / 1e6 / steps, n))
rows.sort(key=lambda r: -abs(r[0]))
print(f"kernel time per step: A {sum(v[1] for v in a.values()) / 1e6 / steps:.2f} ms in {sum(v[0] for v in a.values()) / steps:.0f} launches, "
      f"B {sum(v[1] for v in b.values()) / 1e6 / steps:.2f} ms in {sum(v[0] for v in b.values()) / steps:.0f} launches")
print("  delta ms  delta launches    A ms     B ms   kernel")
for d, k, ta, tb, n in rows[:30]:
    print(f"{d:+9.3f} {k:+10.1f} {ta:10.3f} {tb:8.3f}   {n}")
