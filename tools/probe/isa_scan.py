#!/usr/bin/env python3
"""Memory-pipelining pattern of the kernels of one HIP source: compiles it to gfx950 ISA and prints, per kernel, the sequence of
  L  global / buffer load        S  global store          D  LDS write        R  LDS read
  M  MFMA (runs compressed: M12)  |  s_barrier             w<n>  s_waitcnt vmcnt(n)      B  branch
so that `L w0 D | L w0 D` (a load waited for right where it was issued: no prefetch; hipcc does this to every load that sits under
a branch, even a wave-uniform one) stands out against `L L L ... w2 D`.  Found the epilogue chains of round 4 (store_tile_rows).
usage: python tools/probe/isa_scan.py zs3_amd/csrc/conv_igemm.hip [kernel-name-substring] [max-tokens]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 400
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "zs3_amd", "csrc"),
                "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::|void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    if want not in name:
        continue
    toks = []
    for line in m.group(2).splitlines():
        op = line.strip().split(" ")[0]
        t = None
        if op.startswith(("global_load", "buffer_load")):
            t = "L"
        elif op.startswith(("global_store", "buffer_store")):
            t = "S"
        elif op.startswith("ds_write") or op.startswith("ds_store"):
            t = "D"
        elif op.startswith("ds_read") or op.startswith("ds_load"):
            t = "R"
        elif op.startswith("v_mfma"):
            t = "M"
        elif op == "s_barrier":
            t = "|"
        elif op.startswith(("s_cbranch", "s_branch")):
            t = "B"
        elif op == "s_waitcnt":
            v = re.search(r"vmcnt\((\d+)\)", line)
            if v:
                t = "w" + v.group(1)
        if t is None:
            continue
        if toks and toks[-1][0] == t and t in "MLSDR":
            toks[-1][1] += 1
        else:
            toks.append([t, 1])
    seq = " ".join(t if n == 1 else f"{t}{n}" for t, n in toks)
    print(f"== {name}\n{seq[:limit * 4]}\n")
