"""Builds libzs3hip.so (all HIP kernels, gfx950 only) in-tree with hipcc.  `python -m zs3_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libzs3hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC, "-Wno-unused-result"]


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src, *extra))


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "zs3hip.h")]
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s[:-4] + ".o")
        if force or _newer(src, obj, headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        r = subprocess.run([HIPCC, *FLAGS, "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s[:-4] + ".o") for s in srcs]
    if jobs or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print(f"[zs3_amd.build] {LIB} ({len(jobs)} of {len(srcs)} sources recompiled)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
