"""Per-wave cycle breakdown of the LDS-DMA conv kernel (needs a -DZS3_CONV_TIMING build: ZS3_LIB=...variants/libzs3hip_timing.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
from zs3_amd._lib import lib, P, I, F, stream, check
dev = torch.device("cuda:0")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 31
for (h, ci, co, k) in ((129, 256, 256, 3), (33, 1024, 256, 1), (33, 256, 256, 3), (33, 256, 1024, 1)):
    x = torch.randn(16, h, h, ci, device=dev); wt = torch.randn(co, ci, k, k, device=dev) * 0.02; wp = ops.prep_weight(wt)
    dbg = torch.zeros(96, dtype=torch.int64, device=dev)
    y0, _ = ops.conv2d_fwd(x, wp, 1, k // 2, 1, tile_cfg=cfg)
    n, hh, ww, _ = x.shape
    for rep in range(2):
        check(lib().zs3_conv_igemm(P(x), P(wp.f_pk), P(y0), None, None, P(dbg), None, I(n), I(hh), I(ww), I(hh), I(ww), I(wp.cin_pad), I(ci), I(ci), I(k), I(k), I(1), I(k // 2), I(k // 2), I(1), I(co), I(co), I(0), I(99), F(0.2), I(0), I(0), I(3), I(cfg), P(ops.zero_page(dev)), I(0), stream()), "dbg")
    torch.cuda.synchronize()
    t = dbg.cpu().view(-1, 3)[:8].double()
    KT = k * k * ci // 32
    print(f"{h}^2 {ci}->{co} k{k}: KT={KT} cycles per K step (block 0)")
    tt = dbg.cpu()[24:27].double()
    print(f"   block 0: prologue {tt[0]:.0f} cycles, K loop {tt[2]:.0f}, epilogue {tt[1]:.0f}  (act=99 epilogue: no activation/stats)")
    for w in (0, 3, 4, 7):
        if w < 4:
            print(f"   wave {w} consumer: compute {t[w,0]/KT:.0f}, barrier wait {t[w,2]/KT:.0f}")
        else:
            print(f"   wave {w} producer: issue {t[w,0]/KT:.0f}, vmcnt wait {t[w,1]/KT:.0f}, barrier wait {t[w,2]/KT:.0f}")
