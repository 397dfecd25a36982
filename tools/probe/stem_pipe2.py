import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from zs3_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (n, h, ci, co, k, s, pad) in [(1, 65, 32, 64, 3, 1, 1), (1, 65, 32, 64, 1, 1, 0), (1, 65, 64, 64, 3, 2, 1), (2, 33, 32, 128, 3, 2, 1)]:
    x = torch.randn(n, h, h, ci, generator=g).to(dev)
    w = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).to(dev)
    wp = ops.prep_weight(w)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), stride=s, padding=pad).permute(0, 2, 3, 1)
    for cfg in (14, 11, 4):
        y, _ = ops.conv2d_fwd(x, wp, s, pad, 1, tile_cfg=cfg)
        print((n, h, ci, co, k, s), "cfg", cfg, "rel err", ((y.double() - ref).abs().max() / ref.abs().max()).item())
