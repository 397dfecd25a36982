"""Small-K, wide-N 1x1 layers with COLD operands (a ring of buffers larger than the 256 MB Infinity Cache), the way they run
inside the training step: conv_bench.py re-uses one input and finds it cache-resident.  usage: cold_bench.py [cfgs]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
dev = torch.device("cuda:0")
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "11,14,31").split(",")]
B = 16
for (h, ci, co) in ((129, 64, 256), (129, 256, 64), (65, 128, 512), (33, 256, 1024), (33, 1024, 256)):
    m = B * h * h
    nbuf = max(2, int(1.2e9 / (m * (ci + co) * 4)))
    xs = [torch.randn(B, h, h, ci, device=dev) for _ in range(nbuf)]
    ys = [torch.empty(B, h, h, co, device=dev) for _ in range(nbuf)]
    wp = ops.prep_weight(torch.randn(co, ci, 1, 1, device=dev) * 0.05)
    line = f"{h:3d}^2 {ci:4d}->{co:4d} ({m * (ci + co) * 4 / 1e6:.0f} MB in+out, {nbuf} buffers): "
    for c in cfgs:
        for i in range(nbuf): ops.conv2d_fwd(xs[i], wp, 1, 0, 1, tile_cfg=c, want_stats=True, out=ys[i])
        torch.cuda.synchronize(); t = time.perf_counter()
        reps = 3
        for _ in range(reps):
            for i in range(nbuf): ops.conv2d_fwd(xs[i], wp, 1, 0, 1, tile_cfg=c, want_stats=True, out=ys[i])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / (reps * nbuf)
        line += f" cfg{c}: {dt * 1e6:7.1f} us {m * (ci + co) * 4 / dt / 1e12:5.2f} TB/s |"
    print(line)
