#!/usr/bin/env python3
"""Isolated bandwidth of the elementwise BatchNorm passes at the network's big shapes (bn_act_bwd, affine_act, colstats<1> =
bn_bwd_stats): bytes moved / HIP-event time, 20 launches back to back.  In the step these kernels share the chip with the
weight-gradient streams; this is what they do alone."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from zs3_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(16, 257, 257, 64), (16, 129, 129, 256), (16, 129, 129, 64), (16, 65, 65, 512), (16, 65, 65, 128), (16, 33, 33, 1024), (16, 33, 33, 256),
          (16, 33, 33, 2048)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


for shp in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    dA = torch.randn(shp, device=dev, generator=g)
    y = torch.randn(shp, device=dev, generator=g)
    c = shp[-1]
    mean, istd, gamma = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev) + 0.5
    c1, c2 = torch.randn(c, device=dev) * 0.01, torch.randn(c, device=dev) * 0.01
    sc, sh = gamma * istd, -mean * gamma * istd
    nbytes = dA.numel() * 4
    dy = torch.empty_like(dA)
    t_bwd = timed(lambda: ops.bn_act_bwd(dA, None, y, mean, istd, gamma, c1, c2, dy=dy, act=1, mask_scale=sc, mask_shift=sh))
    out = torch.empty_like(dA)
    t_aff = timed(lambda: ops.affine_act(y, sc, sh, out=out, act=1))
    t_st = timed(lambda: ops.bn_bwd_stats(dA, None, y, mean, istd, sc, sh))
    print(f"{str(shp):24s} {nbytes / 1e6:7.1f} MB/tensor  bn_act_bwd {t_bwd:7.1f} us = {3 * nbytes / t_bwd / 1e6:5.2f} TB/s   affine_act {t_aff:7.1f} us = "
          f"{2 * nbytes / t_aff / 1e6:5.2f} TB/s   bn_bwd_stats {t_st:7.1f} us = {2 * nbytes / t_st / 1e6:5.2f} TB/s")
