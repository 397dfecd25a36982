import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
dev = torch.device("cuda:0")
cfgs = [int(c) for c in sys.argv[1].split(",")]
for (n,h,ci,co,k,s,d) in [(16,33,256,256,3,1,1),(4,65,256,256,3,2,1),(2,33,256,256,3,1,1),(3,17,64,256,1,1,1),(2,65,128,128,3,2,1),(2,17,2048,256,3,1,18),(1,67,304,256,3,1,1),(2,40,256,21,1,1,1),(5,1,2048,256,1,1,1),(16,33,1024,256,1,1,1)]:
    x = torch.randn(n,h,h,ci,device=dev); wt = torch.randn(co,ci,k,k,device=dev)*0.05; wp = ops.prep_weight(wt); pad = d*(k//2)
    y0, s0 = ops.conv2d_fwd(x, wp, s, pad, d, tile_cfg=1, want_stats=True)
    res = torch.randn_like(y0); sc = torch.rand(co, device=dev); sh = torch.randn(co, device=dev)
    z0,_ = ops.conv2d_fwd(x, wp, s, pad, d, tile_cfg=1, scale=sc, shift=sh, res=res, act=1)
    dy = torch.randn(n, y0.shape[1], y0.shape[2], (co+7)//8*8, device=dev)[..., :co]
    d0 = ops.conv2d_dgrad(dy, wp, (h,h), s, pad, d, tile_cfg=1)
    for c in cfgs:
      try:
        y1, s1 = ops.conv2d_fwd(x, wp, s, pad, d, tile_cfg=c, want_stats=True)
        z1,_ = ops.conv2d_fwd(x, wp, s, pad, d, tile_cfg=c, scale=sc, shift=sh, res=res, act=1)
        d1 = ops.conv2d_dgrad(dy, wp, (h,h), s, pad, d, tile_cfg=c)
        torch.cuda.synchronize()
      except RuntimeError as e:
        print((n,h,ci,co,k,s,d), "cfg", c, "n/a:", str(e)[:60]); continue
      print((n,h,ci,co,k,s,d), "cfg", c, "dy", (y1-y0).abs().max().item(), "dstat", (s1.sum(0)-s0.sum(0)).abs().max().item()/s0.sum(0).abs().max().item(), "dz", (z1-z0).abs().max().item(), "ddx", (d1-d0).abs().max().item())
