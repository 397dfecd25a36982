"""Per-wave cycle split of one workgroup of the persistent pointwise kernel (conv_pw.hip).  Needs a -DZS3_CONV_TIMING build:
ZS3_VARIANT_SRC=conv_pw tools/probe/build_variant.sh pwtiming -DZS3_CONV_TIMING; ZS3_LIB=zs3_amd/lib/variants/libzs3hip_pwtiming.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
from zs3_amd._lib import lib
dev = torch.device("cuda:0")
dbg = torch.zeros(256, dtype=torch.int64, device=dev)
fn = lib().zs3_conv_pw_timing
fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
fn(dbg.data_ptr())
for (h, ci, co) in ((33, 256, 1024), (129, 64, 256), (33, 512, 2048), (65, 128, 512)):
    for cfg in (52, 51):
        x = torch.randn(16, h, h, ci, device=dev); wt = torch.randn(co, ci, 1, 1, device=dev) * 0.02
        wp = ops.prep_weight(wt, f16_forward=True)
        for blk in (0, 100):
            dbg.zero_(); dbg[0] = blk
            for rep in range(3):
                y, st = ops.conv2d_fwd(x, wp, 1, 0, 1, want_stats=True, tile_cfg=cfg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y, st = ops.conv2d_fwd(x, wp, 1, 0, 1, want_stats=True, tile_cfg=cfg); e1.record(); torch.cuda.synchronize()
            t = dbg.cpu()[8:8 + 40].view(8, 5).double()
            bm = 128 if cfg == 52 else 256
            tiles = ((16 * h * h + bm - 1) // bm) * (co // 128)
            nk = ci // 32
            print(f"cfg{cfg} {h}^2 {ci}->{co}: {tiles} tiles, {nk} K steps each, kernel {e0.elapsed_time(e1)*1e3:.1f} us (events); block {blk}: cycles (100 MHz ticks?)")
            for w in (0, 3):
                print(f"   wave {w} consumer: total {t[w,4]:8.0f}  before first step {t[w,3]:7.0f}  MFMA sub-steps {t[w,0]:8.0f}  epilogues {t[w,1]:8.0f}  barriers {t[w,2]:8.0f}")
            for w in (4, 7):
                print(f"   wave {w} producer: total {t[w,4]:8.0f}  prologue {t[w,3]:7.0f}  load wait {t[w,0]:8.0f}  split+write+issue {t[w,1]:8.0f}  barriers {t[w,2]:8.0f}")
