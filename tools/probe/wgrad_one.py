import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
dev = torch.device("cuda:0")
h, ci, co, k, d = (int(a) for a in sys.argv[1:6])
x = torch.randn(16, h, h, ci, device=dev); dy = torch.randn(16, h, h, co, device=dev); pad = d * (k // 2)
fn = lambda: ops.conv2d_wgrad(dy, x, co, ci, k, k, 1, pad, pad, d)
for _ in range(2): fn()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): fn()
torch.cuda.synchronize(); t = (time.perf_counter() - t) / 5
print(f"split={os.environ.get('ZS3_WGRAD_SPLIT')} {h}^2 {ci}->{co} k{k} d{d}: {t*1e6:.1f} us {2.0*16*h*h*co*ci*k*k/t/1e12:.1f} TF")
