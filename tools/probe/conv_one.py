import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
dev = torch.device("cuda:0")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h, ci, co, k = (int(a) for a in (sys.argv[2:6] if len(sys.argv) > 5 else (129, 256, 256, 3)))
x = torch.randn(16, h, h, ci, device=dev); wt = torch.randn(co, ci, k, k, device=dev) * 0.02
wp = ops.prep_weight(wt)
for _ in range(3):
    ops.conv2d_fwd(x, wp, 1, k // 2, 1, tile_cfg=cfg, want_stats=True)
torch.cuda.synchronize()
