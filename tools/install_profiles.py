"""gpurun_out/prof/* (tools/refresh_profiles.sh, run on the GPU box) -> profiles/rNN_*: adds the header (round, command, the bench
line of the traced run) the committed summaries carry.  usage: python tools/install_profiles.py 3"""
import json, os, sys
rnd = int(sys.argv[1])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof"), os.path.join(root, "profiles")
base = "python bench.py --no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --host-steps 0 --script-steps 0"
runs = {"sup": ("supervised", base + " --steps 5 --warmup 2 --gmmn-steps 0", "11 steps in the trace: 4 that settle and record the launch plan, 2 warm-up, 5 timed (4 of them replayed from the plan)"),
        "bf16": ("supervised_bf16", base + " --steps 5 --warmup 2 --gmmn-steps 0 --dtype bf16", "11 steps in the trace: 4 that settle and record the launch plan, 2 warm-up, 5 timed (4 of them replayed from the plan)"),
        "gmmn": ("gmmn", base + " --workload gmmn --steps 4 --warmup 2 --no-roofline", "6 steps in the trace: 2 warm-up + 4 timed; the feature pass is recorded on its third call and replayed afterwards")}
for key, (name, cmd, note) in runs.items():
    line = json.loads(open(os.path.join(src, f"bench_{key}.json")).read())
    head = f"# r{rnd}_{name}_kernel_stats\n\nRound {rnd}, 1x MI355X, B=16, 513x513, 21 classes, synthetic data; `rocprofv3 --kernel-trace --stats -- {cmd}` " \
           f"(tools/refresh_profiles.sh; {note}; the tracer slows the host, so step time under the tracer is not the bench number).\n\n"
    head += f"bench line of the traced run: {line['ms_per_step']:.2f} ms per step under the tracer"
    r = line.get("roofline")
    if r:
        head += f"; dominant kernel {r['kernel']}: {r['achieved']:.1f} TF = {r['frac']:.3f} of 2.5 PF (HIP events, in-step)"
    body = open(os.path.join(src, f"kt_{key}.md")).read()
    open(os.path.join(dst, f"r{rnd}_{name}_kernel_stats.md"), "w").write(head + "\n\n" + body)
open(os.path.join(dst, f"r{rnd}_pmc_traffic.json"), "w").write(open(os.path.join(src, "pmc_traffic.json")).read())
if os.path.exists(os.path.join(src, "pmc_traffic_bf16.json")):     # the 2-byte mode's passes (round 4)
    db = json.load(open(os.path.join(src, "pmc_traffic_bf16.json")))
    fam16 = [v for k, v in db["kernels"].items() if k.startswith("conv_halo_kernel<1,")]
    n16 = sum(v["launches"] for v in fam16)
    if n16:
        db["conv_halo_family"] = {"launches": n16,
                                  "read_bytes_per_launch": sum(v["read_bytes_per_launch"] * v["launches"] for v in fam16) / n16,
                                  "write_bytes_per_launch": sum(v["write_bytes_per_launch"] * v["launches"] for v in fam16) / n16}
    db["command"] = ("rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE, a separate pass) --kernel-trace -- " + base +
                     f" --steps 2 --warmup 1 --gmmn-steps 0 --no-roofline --dtype bf16 (tools/refresh_profiles.sh, round {rnd})")
    json.dump(db, open(os.path.join(dst, f"r{rnd}_pmc_traffic_bf16.json"), "w"), indent=1)
old = open(os.path.join(dst, f"r{rnd}_pmc_mfma.md")).read() if os.path.exists(os.path.join(dst, f"r{rnd}_pmc_mfma.md")) else ""
head = old.split("\ncounters:")[0] if "\ncounters:" in old else f"# r{rnd}_pmc_mfma\n"
new = open(os.path.join(src, "pmc_mfma.md")).read()
open(os.path.join(dst, f"r{rnd}_pmc_mfma.md"), "w").write(head + "\n" + new[new.index("counters:"):] if "counters:" in new else head + "\n" + new)
print("installed", sorted(f for f in os.listdir(dst) if f.startswith(f"r{rnd}_")))
# the strip kernel's instantiations as one family (what bench.py's roofline.traffic quotes), and the command
p = os.path.join(dst, f"r{rnd}_pmc_traffic.json")
d = json.load(open(p))
fam = [v for k, v in d["kernels"].items() if k.startswith("conv_halo_kernel<3,") or k.startswith("conv_halo_kernel<4,")]   # bf16x3 (dgrad) + f16x3 (forward)
n = sum(v["launches"] for v in fam)
d["conv_halo_family"] = {"launches": n, "read_bytes_per_launch": sum(v["read_bytes_per_launch"] * v["launches"] for v in fam) / n,
                         "write_bytes_per_launch": sum(v["write_bytes_per_launch"] * v["launches"] for v in fam) / n}
d["command"] = ("rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE, a separate pass) --kernel-trace -- " + base +
                f" --steps 2 --warmup 1 --gmmn-steps 0 --no-roofline (tools/refresh_profiles.sh, round {rnd})")
json.dump(d, open(p, "w"), indent=1)
