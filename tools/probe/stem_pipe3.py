import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F, torch.nn.functional as TF
from zs3_amd import ops
from zs3_amd._lib import I, P, check, lib, stream
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for hw in (33, 65, 513):
    image = torch.randn(1, 3, hw, hw, generator=g).to(dev)
    wt = (torch.randn(64, 3, 7, 7, generator=g) / 12.0).to(dev)
    n, c, h, w = image.shape
    ho, wo = ops.conv_out_size(h, 7, 2, 3, 1), ops.conv_out_size(w, 7, 2, 3, 1)
    wp_ = max(w + 7, 2 * (wo - 1) + 8)
    xp = torch.empty((n, h, wp_, 4), dtype=torch.float32, device=dev)
    check(lib().zs3_nchw3_to_nhwc4(P(image), P(xp), I(n), I(h), I(w), I(wp_), I(3), stream()), "x")
    w_eff = TF.pad(wt.permute(0, 2, 3, 1), (0, 1, 0, 1)).reshape(64, 7, 1, 32).permute(0, 3, 1, 2)
    wpl = ops.prep_weight(w_eff)
    geom = dict(ho=ho, wo=wo, cin_pad=32, cin_valid=32, kh=7, kw=1, stride=2, pad_h=3, pad_w=0, dil=1, ncols=64)
    ref = F.conv2d(image.double(), wt.double(), stride=2, padding=3).permute(0, 2, 3, 1)
    for cfg in (0, 14, 11, 4):
        y, st = ops.conv_igemm(xp, wpl.f_pk, want_stats=True, tile_cfg=cfg, **geom)
        print(hw, "cfg", cfg, "rel err", ((y.double() - ref).abs().max() / ref.abs().max()).item(), "sum|y|", float(y.abs().sum()))
