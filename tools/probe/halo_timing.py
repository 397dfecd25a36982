"""Per-wave cycle breakdown of the strip-resident conv kernel's K loop (block 0).  Needs a -DZS3_CONV_TIMING build:
ZS3_VARIANT_SRC=conv_halo tools/probe/build_variant.sh timing -DZS3_CONV_TIMING; ZS3_LIB=zs3_amd/lib/variants/libzs3hip_timing.so"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
from zs3_amd._lib import lib, P, I, F, stream, check
dev = torch.device("cuda:0")
for cfg in (41, 42):
    for (h, ci, co, d) in ((33, 256, 256, 1), (129, 256, 256, 1), (33, 512, 512, 4), (33, 2048, 256, 6)):
        x = torch.randn(16, h, h, ci, device=dev); wt = torch.randn(co, ci, 3, 3, device=dev) * 0.02; wp = ops.prep_weight(wt)
        if not ops.halo_ok(x.shape, h, h, wp.cin_pad, ci, ci, 3, 3, 1, d, d, d, False, 3, cfg):
            continue
        dbg = torch.zeros(4096, dtype=torch.int64, device=dev)
        y0 = torch.empty(16, h, h, co, device=dev)
        for rep in range(2):
            check(lib().zs3_conv_igemm(P(x), P(wp.f_pk), P(y0), None, None, P(dbg), None, I(16), I(h), I(h), I(h), I(h), I(wp.cin_pad), I(ci), I(ci), I(3), I(3), I(1), I(d), I(d), I(d), I(co), I(co), I(0), I(99), F(0.2), I(0), I(0), I(3), I(cfg), P(ops.zero_page(dev)), I(0), stream()), "dbg")
        torch.cuda.synchronize()
        t = dbg.cpu()[:24].view(8, 3).double()
        ns = 9 * ((ci + 15) // 16)
        print(f"cfg{cfg} {h}^2 {ci}->{co} d{d}: {ns} K16 steps; cycles per step in block 0")
        for w in (0, 3):
            print(f"   wave {w} consumer:        work {t[w,0]/ns:6.0f}  barrier {t[w,2]/ns:6.0f}")
        for w in (4, 5):
            print(f"   wave {w} weight producer: issue {t[w,0]/ns:6.0f}  vmcnt wait {t[w,1]/ns:6.0f}  barrier {t[w,2]/ns:6.0f}")
        for w in (6, 7):
            print(f"   wave {w} strip producer:  write+load {t[w,0]/ns:6.0f}  barrier {t[w,2]/ns:6.0f}")
