"""M=128 row-GEMMs of the GMMN generator: which tile config is fastest (latency-bound regime)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=100):
    """GPU time per launch of a dependent chain: `iters` launches captured in one hipGraph, replayed 5 times"""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters): fn()
        g.replay(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / (5 * iters) * 1e6
for (m, k, n) in ((128, 600, 256), (128, 256, 256)):
    x = torch.randn(1, 1, m, (k + 7) // 8 * 8, device=dev)[..., :k]
    w = torch.randn(n, k, device=dev) * 0.05
    wp = ops.prep_weight(w)
    dy = torch.randn(1, 1, m, n, device=dev)
    dx_in = torch.randn(1, 1, m, k, device=dev)
    line = f"M={m} K={k} N={n}: "
    for cfg in (14, 13, 12, 11, 4, 3, 2, 1):
        t = timeit(lambda: ops.conv2d_fwd(x, wp, tile_cfg=cfg))
        td = timeit(lambda: ops.conv2d_dgrad(dy, wp, (1, m), tile_cfg=cfg))
        line += f" cfg{cfg}: fwd {t:.1f} dgrad {td:.1f} |"
    print(line)
    tw = timeit(lambda: ops.conv2d_wgrad(dy, dx_in, n, k, 1, 1))
    print(f"   wgrad {tw:.1f} us (host-paced launches: includes ~python overhead if the GPU is faster)")
