"""wgrad kernel check against an fp64 torch reference (run with ZS3_WGRAD_KERNEL=1|2 to force a kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from zs3_amd import ops
dev = torch.device("cuda:0")
for (n, h, w, ci, co, k, s, d) in [(2, 33, 33, 256, 256, 3, 1, 1), (4, 33, 31, 256, 512, 1, 1, 1), (2, 35, 33, 512, 256, 3, 2, 1),
                                   (1, 17, 19, 512, 256, 3, 1, 6), (16, 33, 33, 1024, 256, 1, 1, 1), (3, 20, 16, 256, 256, 3, 1, 2),
                                   (2, 65, 65, 256, 512, 1, 2, 1), (1, 16, 16, 256, 256, 1, 1, 1)]:
    g = torch.Generator().manual_seed(h + ci)
    x = torch.randn(n, ci, h, w, generator=g); pad = d * (k // 2)
    ho, wo = ops.conv_out_size(h, k, s, pad, d), ops.conv_out_size(w, k, s, pad, d)
    dy = torch.randn(n, co, ho, wo, generator=g)
    wr = torch.zeros(co, ci, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wr, stride=s, padding=pad, dilation=d).backward(dy.double())
    dw = ops.conv2d_wgrad(dy.to(dev).permute(0, 2, 3, 1).contiguous(), x.to(dev).permute(0, 2, 3, 1).contiguous(), co, ci, k, k, s, pad, pad, d)
    torch.cuda.synchronize()
    err = ((dw.permute(0, 3, 1, 2).double().cpu() - wr.grad).abs().max() / wr.grad.abs().max()).item()
    print((n, h, w, ci, co, k, s, d), "rel err", err, "OK" if err < 5e-5 else "FAIL")
