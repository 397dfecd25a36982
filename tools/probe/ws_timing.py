import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
dev = torch.device("cuda:0")
for (h, ci, co, k) in ((129,256,256,3),(33,1024,256,1)):
    x = torch.randn(16,h,h,ci,device=dev); wt = torch.randn(co,ci,k,k,device=dev)*0.02; wp = ops.prep_weight(wt)
    dbg = torch.zeros(96, dtype=torch.int64, device=dev)
    y0,_ = ops.conv2d_fwd(x, wp, 1, k//2, 1, tile_cfg=21)
    dbgf = dbg.view(torch.float32).view(1,1,1,-1)
    # act=99 -> timers written to res pointer
    from zs3_amd._lib import lib, P, I, F, stream, check
    n,hh,ww,_ = x.shape
    for abl in (0,):
      for rep in range(2):
        check(lib().zs3_conv_igemm(P(x), P(wp.f_pk), P(y0), None, None, P(dbg), None, I(n), I(hh), I(ww), I(hh), I(ww), I(wp.cin_pad), I(ci), I(ci), I(k), I(k), I(1), I(k//2), I(k//2), I(1), I(co), I(co), I(0), I(99), F(0.2), I(abl), I(0), I(3), I(21), P(ops.zero_page(dev)), stream()), "dbg")
      torch.cuda.synchronize()
      t = dbg.cpu().view(-1, 3)[:8]
      KT = k*k*ci//32
      print(f"{h}^2 {ci}->{co} k{k}: KT={KT} ablate={abl} (2: no global loads, 4: no cvt/ds_write)")
      for w in (0, 1, 4, 5):
        print(f"   wave {w} ({'consumer' if w<4 else 'producer'}): work {t[w,0].item()/KT:.0f} ticks/kstep, barrier wait {t[w,1].item()/KT:.0f}, store part {t[w,2].item()/KT:.0f} ticks/kstep")
