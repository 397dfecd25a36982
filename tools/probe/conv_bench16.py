"""Per-layer table of the conv kernels at B=16, 513x513 with the epilogues the training step really runs, in either storage form.
usage: [ZS3_STORAGE=bf16|fp32] [ZS3_PREC=1|3] conv_bench16.py cfgs(comma) fwd|dgrad|dgrad_epi   (dgrad_epi: lazily masked skip
gradient + fused BatchNorm-backward sums with mask bits: the epilogue of conv1's data gradient in an identity block; for
columns < 128 or 3x3 layers: BatchNorm-backward sums with the mask from y only)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
dev = torch.device("cuda:0")
from conv_bench import SHAPES  # noqa: E402  (count, H, Cin, Cout, k, stride, dil)
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
bf = os.environ.get("ZS3_STORAGE", "bf16") == "bf16"
if bf:
    ops.set_storage(torch.bfloat16)
else:
    ops.PREC_DEFAULT = int(os.environ.get("ZS3_PREC", "3"))
dt = torch.bfloat16 if bf else torch.float32
B = 16
def timeit(fn, iters=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / iters
tot = {c: 0.0 for c in cfgs}; totfl = 0.0
for (cnt, h, ci, co, k, s, d) in SHAPES:
    x = torch.randn(B, h, h, ci, device=dev).to(dt); wt = torch.randn(co, ci, k, k, device=dev) * 0.02
    wp = ops.prep_weight(wt); pad = d * (k // 2)
    ho = ops.conv_out_size(h, k, s, pad, d)
    fl = 2.0 * B * ho * ho * co * ci * k * k
    dy = torch.randn(B, ho, ho, (co + 7) // 8 * 8, device=dev).to(dt)[..., :co]
    line = f"{cnt:2d}x {h:3d}^2 {ci:4d}->{co:4d} k{k} s{s} d{d:2d}: "
    if mode == "dgrad_epi":
        yprev = torch.randn(B, h, h, ci, device=dev).to(dt)
        mean, istd = torch.randn(ci, device=dev) * 0.1, torch.rand(ci, device=dev) + 0.5
        msc, msh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.3
        bits = torch.randint(0, 256, (B * h * h * ci // 4,), device=dev, dtype=torch.uint8) if ci % 4 == 0 else None
        skip = torch.randn(B, h, h, ci, device=dev).to(dt)
        full = k == 1 and s == 1 and ci >= 128 and bits is not None
    for c in cfgs:
        try:
            if mode == "fwd": t = timeit(lambda: ops.conv2d_fwd(x, wp, s, pad, d, tile_cfg=c, want_stats=True))
            elif mode == "dgrad": t = timeit(lambda: ops.conv2d_dgrad(dy, wp, (h, h), s, pad, d, tile_cfg=c, out_dtype=dt))
            elif full: t = timeit(lambda: ops.conv2d_dgrad(dy, wp, (h, h), s, pad, d, tile_cfg=c, out=skip, res=skip, res_mask_bits=bits,
                                                           bn_bwd=(yprev, mean, istd, None, None, bits)))
            elif ci % 4 == 0: t = timeit(lambda: ops.conv2d_dgrad(dy, wp, (h, h), s, pad, d, tile_cfg=c, out_dtype=dt,
                                                                   bn_bwd=(yprev, mean, istd, msc, msh, None)))
            else: t = timeit(lambda: ops.conv2d_dgrad(dy, wp, (h, h), s, pad, d, tile_cfg=c, out_dtype=dt))
        except Exception as e:
            line += f" cfg{c}: {type(e).__name__:>12s} |"; continue
        tot[c] += cnt * t
        line += f" cfg{c}: {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF |"
    totfl += cnt * fl
    print(line)
print("TOTAL per pass:", {c: f"{v*1e3:.2f} ms ({totfl/v/1e12:.0f} TF)" for c, v in tot.items()})
