"""BN-apply in the consumer's operand path, per layer: conv(a) + affine_act(y) against conv(y, in_affine) -- forward and weight gradient."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from zs3_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / iters * 1e6
for (h, ci, co, k, d) in ((33, 256, 256, 3, 1), (33, 256, 1024, 1, 1), (33, 512, 512, 3, 2), (33, 512, 2048, 1, 1), (129, 64, 256, 1, 1), (65, 128, 512, 1, 1)):
    y = torch.randn(16, h, h, ci, device=dev); sc = torch.rand(ci, device=dev) + 0.5; sh = torch.randn(ci, device=dev)
    wp = ops.prep_weight(torch.randn(co, ci, k, k, device=dev) * 0.02); pad = d * (k // 2)
    a = ops.affine_act(y, sc, sh, act=1)
    dy = torch.randn(16, h, h, co, device=dev)
    t_aff = timeit(lambda: ops.affine_act(y, sc, sh, act=1, out=a))
    t_f0 = timeit(lambda: ops.conv2d_fwd(a, wp, 1, pad, d, want_stats=True))
    t_f1 = timeit(lambda: ops.conv2d_fwd(y, wp, 1, pad, d, want_stats=True, in_affine=(sc, sh)))
    t_w0 = timeit(lambda: ops.conv2d_wgrad(dy, a, co, ci, k, k, 1, pad, pad, d))
    t_w1 = timeit(lambda: ops.conv2d_wgrad(dy, y, co, ci, k, k, 1, pad, pad, d, x_affine=(sc, sh)))
    print(f"{h:3d}^2 {ci:4d}->{co:4d} k{k} d{d}: affine_act {t_aff:6.1f} us | fwd {t_f0:6.1f} -> {t_f1:6.1f} us | wgrad {t_w0:6.1f} -> {t_w1:6.1f} us")
