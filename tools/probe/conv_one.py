"""One conv layer, a few launches (for rocprofv3 --pmc runs).  usage: conv_one.py cfg H Cin Cout k [dil] [mode]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
dev = torch.device("cuda:0")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h, ci, co, k = (int(a) for a in (sys.argv[2:6] if len(sys.argv) > 5 else (129, 256, 256, 3)))
d = int(sys.argv[6]) if len(sys.argv) > 6 else 1
mode = sys.argv[7] if len(sys.argv) > 7 else "fwd"
x = torch.randn(16, h, h, ci, device=dev); wt = torch.randn(co, ci, k, k, device=dev) * 0.02
wp = ops.prep_weight(wt)
pad = d * (k // 2)
dy = torch.randn(16, h, h, co, device=dev)
for _ in range(3):
    if mode == "fwd":
        ops.conv2d_fwd(x, wp, 1, pad, d, tile_cfg=cfg, want_stats=True)
    elif mode == "dgrad":
        ops.conv2d_dgrad(dy, wp, (h, h), 1, pad, d, tile_cfg=cfg)
    else:
        ops.conv2d_wgrad(dy, x, co, ci, k, k, 1, pad, pad, d)
torch.cuda.synchronize()
