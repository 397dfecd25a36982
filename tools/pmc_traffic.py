#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) into per-kernel HBM bytes per launch.
usage: pmc_traffic.py <dir with pmcb_FETCH_SIZE/ and pmcb_WRITE_SIZE/> <steps> > profiles/rNN_pmc_traffic.json

Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is."""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return re.sub(r"\(.*\)$", "", name)[:90]


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main(root, steps):
    f = load(f"{root}/pmcb_FETCH_SIZE/p_counter_collection.csv")
    w = load(f"{root}/pmcb_WRITE_SIZE/p_counter_collection.csv")
    if steps <= 0:      # one optimizer launch per training step: count them (round 6: the bench adds plan-recording steps of its own)
        steps = max(1, sum(v[0] for k, v in f.items() if "sgd_multi" in k))
    out = {"steps_profiled": steps, "note": "HBM bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH correction)",
           "per_step_read_GB": sum(v[1] for v in f.values()) * 2 * 1024 / steps / 1e9,
           "per_step_write_GB": sum(v[1] for v in w.values()) * 1024 / steps / 1e9, "kernels": {}}
    for k in sorted(f, key=lambda k: -(f[k][1] * 2 + w.get(k, [0, 0])[1])):
        n = f[k][0]
        wr = w.get(k, [1, 0.0])
        out["kernels"][k] = {"launches": n, "read_bytes_per_launch": f[k][1] * 2 * 1024 / n,
                             "write_bytes_per_launch": wr[1] * 1024 / max(wr[0], 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
