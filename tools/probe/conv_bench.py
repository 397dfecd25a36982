"""Weighted micro-benchmark of the implicit-GEMM conv over the DeepLabv3+ layer shapes at B=16, 513x513.
usage: conv_bench.py [cfgs comma separated, e.g. 1,11] [fwd|dgrad|wgrad]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
dev = torch.device("cuda:0")
# (count, H, Cin, Cout, k, stride, dil)
SHAPES = [(23, 33, 256, 1024, 1, 1, 1), (22, 33, 1024, 256, 1, 1, 1), (22, 33, 256, 256, 3, 1, 1), (4, 129, 64, 256, 1, 1, 1),
          (4, 65, 128, 512, 1, 1, 1), (3, 129, 64, 64, 3, 1, 1), (3, 65, 512, 128, 1, 1, 1), (3, 65, 128, 128, 3, 1, 1),
          (3, 33, 512, 2048, 1, 1, 1), (2, 129, 256, 64, 1, 1, 1), (2, 33, 2048, 512, 1, 1, 1), (1, 129, 128, 128, 3, 2, 1),
          (1, 129, 256, 512, 1, 2, 1), (1, 65, 256, 256, 3, 2, 1), (1, 65, 512, 1024, 1, 2, 1), (1, 33, 1024, 512, 1, 1, 1),
          (1, 33, 512, 512, 3, 1, 2), (1, 33, 512, 512, 3, 1, 4), (1, 33, 512, 512, 3, 1, 8), (1, 33, 1024, 2048, 1, 1, 1),
          (1, 33, 2048, 256, 1, 1, 1), (1, 33, 2048, 256, 3, 1, 6), (1, 33, 2048, 256, 3, 1, 12), (1, 33, 2048, 256, 3, 1, 18),
          (1, 33, 1280, 256, 1, 1, 1), (1, 129, 256, 48, 1, 1, 1), (1, 129, 304, 256, 3, 1, 1), (1, 129, 256, 256, 3, 1, 1)]
if os.environ.get("ZS3_SHAPES"):   # e.g. ZS3_SHAPES=2,16,26: only these rows of the table
    SHAPES = [SHAPES[int(i)] for i in os.environ["ZS3_SHAPES"].split(",")]
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
ops.PREC_DEFAULT = int(os.environ.get("ZS3_PREC", "3"))   # 1 = plain bf16 products
B = int(os.environ.get("ZS3_B", "16"))
def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / iters
tot = {c: 0.0 for c in cfgs}; totfl = 0.0
for (cnt, h, ci, co, k, s, d) in SHAPES:
    x = torch.randn(B, h, h, ci, device=dev); wt = torch.randn(co, ci, k, k, device=dev) * 0.02
    wp = ops.prep_weight(wt); pad = d * (k // 2)
    if os.environ.get("ZS3_A16") == "1": x = x.bfloat16()      # bf16-stored input (tile_cfg 141 / 142, ZS3_PREC=1)
    ho = ops.conv_out_size(h, k, s, pad, d)
    fl = 2.0 * B * ho * ho * co * ci * k * k
    dy = torch.randn(B, ho, ho, (co + 7) // 8 * 8, device=dev)[..., :co]
    line = f"{cnt:2d}x {h:3d}^2 {ci:4d}->{co:4d} k{k} s{s} d{d:2d}: "
    for c0 in cfgs:
        c = c0
        if mode == "fwd": t = timeit(lambda: ops.conv2d_fwd(x, wp, s, pad, d, tile_cfg=c, want_stats=True))
        elif mode == "dgrad": t = timeit(lambda: ops.conv2d_dgrad(dy, wp, (h, h), s, pad, d, tile_cfg=c))
        else: t = timeit(lambda: ops.conv2d_wgrad(dy, x, co, ci, k, k, s, pad, pad, d))
        tot[c0] += cnt * t
        line += f" cfg{c}: {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF |"
    totfl += cnt * fl
    print(line)
print("TOTAL per pass:", {c: f"{v*1e3:.2f} ms ({totfl/v/1e12:.0f} TF)" for c, v in tot.items()})
