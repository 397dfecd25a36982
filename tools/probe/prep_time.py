import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd import functional as Fz
dev = torch.device("cuda:0")
m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
ws = [p for p in m.parameters() if p.dim() == 4]
for w in ws[1:]:
    Fz.weight_planes(w, need_t=True)
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("refresh_planes (one launch, all conv weights): %.3f ms" % t(lambda: Fz.refresh_planes(*ws[1:])))
