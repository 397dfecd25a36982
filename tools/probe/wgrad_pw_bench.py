"""Pointwise weight-gradient launches of the 513^2, B = 16 step, alone on the GPU: microseconds per launch for the layer shapes
of ResNet-101 layers 3 / 4 (resnet.py:16-28).  ZS3_WGRAD_PW_WIDE=1024 selects the (4, 2) / (2, 4) consumer blocks."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zs3_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
cases = [(16, 33, 33, 256, 1024), (16, 33, 33, 1024, 256), (16, 33, 33, 512, 2048), (16, 33, 33, 2048, 512), (16, 65, 65, 128, 512),
         (16, 65, 65, 512, 128), (16, 33, 33, 256, 256)]
out = []
for (n, h, w, ci, co) in cases:
    x = torch.randn(n, h, w, ci, device=dev)
    dy = torch.randn(n, h, w, co, device=dev)
    for _ in range(5):
        dw = ops.conv2d_wgrad(dy, x, co, ci, 1, 1, 1, 0, 0, 1, prec=3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        dw = ops.conv2d_wgrad(dy, x, co, ci, 1, 1, 1, 0, 0, 1, prec=3)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    ref = dy.reshape(-1, co)[:, :8].double().t() @ x.reshape(-1, ci).double()
    err = ((dw.view(co, ci)[:8].double() - ref).abs().max() / ref.abs().max()).item()
    flops = 2.0 * n * h * w * ci * co
    out.append(f"{ci}->{co}@{h}: {us:.1f} us  {flops / us / 1e6:.0f} TF  err {err:.1e}")
print("WIDE=" + os.environ.get("ZS3_WGRAD_PW_WIDE", "0") + " | " + " | ".join(out))
