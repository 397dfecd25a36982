#!/usr/bin/env python3
"""MFMA-busy evidence from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE ...; csv).
usage: pmc_mfma.py <p_counter_collection.csv> <steps> > profiles/rNN_pmc_mfma.md

Per kernel: launches, GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES, and MFMA-busy = busy / (active * 128): the fraction of
SIMD-cycles in which the matrix pipe was busy (one v_mfma_f32_32x32x16_bf16 = 32 busy cycles, MI355X_MICROARCH.md).
Both counters arrive summed over the 8 XCDs: SQ_VALU_MFMA_BUSY_CYCLES over all 1024 SIMDs (checked against the launches'
MFMA count: 127 LDS-DMA conv launches per step = 4860 GFLOP x 3 products / 32768 flop per MFMA x 32 cycles = 14.2 G busy
cycles per step), GRBM_GUI_ACTIVE over the 8 per-XCD GRBMs (8 x kernel time x clock), so the SIMD-cycles available are
(active / 8) * 1024 = active * 128.  For bf16x3 kernels every algorithmic
product costs three MFMAs, so MFMA-busy ~ 3 x (achieved TF / 2500 TF) x (2.4 GHz / actual clock)."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return re.sub(r"\(.*\)$", "", name)[:80]


def main(path, steps):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
    if steps <= 0:      # one optimizer launch per training step (the bench adds plan-recording steps of its own since round 6)
        steps = max(1, sum(n for k, n in cnt.items() if "sgd_multi" in k))
    names = sorted({c for v in agg.values() for c in v})
    print(f"counters: {', '.join(names)}; {steps} training steps profiled (PMC collection serialises kernels: durations are not step-time)\n")
    # effective clock of a kernel: GRBM_GUI_ACTIVE is summed over the 8 XCDs, so cycles per XCD / kernel duration = the clock the
    # kernel actually ran at (durations from the trace columns of the same CSV; VERDICT r3 #8: the sustained-clock ceiling as a column)
    dur = collections.defaultdict(float)
    seen_d = set()
    for r in csv.DictReader(open(path)):
        key = (r.get("Dispatch_Id"), short(r["Kernel_Name"]))
        if key in seen_d or "Start_Timestamp" not in r:
            continue
        seen_d.add(key)
        dur[key[1]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    print("| kernel | launches | GRBM_GUI_ACTIVE (Mcyc) | effective clock (GHz) | SQ_VALU_MFMA_BUSY_CYCLES (Mcyc) | MFMA-busy | share of all MFMA-busy cycles "
          "| wave cycles: parked (WAIT_ANY) / issue-stalled (WAIT_INST_ANY) / issuing (ACTIVE_INST_ANY) | VALU share of issuing "
          "| LDS bank-conflict cycles per wave cycle |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    tot_busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in agg.values()) or 1.0
    rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0))
    for k, v in rows[:25]:
        act, busy = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        util = busy / (act * 128.0) if act else float("nan")
        wc = v.get("SQ_WAVE_CYCLES", 0.0) or float("nan")
        extra = (f"{100 * v.get('SQ_WAIT_ANY', 0) / wc:.0f} % / {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} % / "
                 f"{100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f} %")
        valu = v.get("SQ_ACTIVE_INST_VALU", 0.0) / (v.get("SQ_ACTIVE_INST_ANY", 0.0) or float("nan"))
        ghz = (act / 8.0) / dur[k] if dur.get(k) else float("nan")     # cycles per XCD / nanoseconds
        print(f"| {k} | {cnt[k]} | {act / 1e6:.1f} | {ghz:.2f} | {busy / 1e6:.1f} | {100 * util:.1f} % | {100 * busy / tot_busy:.1f} % | {extra} | "
              f"{100 * valu:.0f} % | {v.get('SQ_LDS_BANK_CONFLICT', 0) / wc:.3f} |")
    act = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in agg.values())
    print(f"\nall kernels: MFMA-busy {100 * tot_busy / (act * 128.0):.1f} % of the GPU-active SIMD-cycles")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
