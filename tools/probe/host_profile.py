"""cProfile of the host side of the supervised step (where the ~38 ms of issue time per step go).  usage: host_profile.py [steps]"""
import sys, os, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.utils.loss import SegmentationLosses
from zs3_amd.optim import SGD
import os
from zs3_amd import ops
if os.environ.get("ZS3_STORAGE") == "bf16":
    ops.set_storage(torch.bfloat16)
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.manual_seed(1)
m = DeepLab(num_classes=21, pretrained=False, sync_bn=False).to(dev).train()
opt = SGD([{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}], momentum=0.9, weight_decay=5e-4)
crit = SegmentationLosses(cuda=True).build_loss("ce")
x = torch.randn(16, 3, 513, 513, device=dev); y = torch.randint(0, 21, (16, 513, 513), device=dev).float()
def step():
    opt.zero_grad(); out = m(x); loss = crit(out, y); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(steps): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)

# ---- the backward pass runs in the autograd engine's device thread, which the profile above does not see: a second profiler
# is switched on INSIDE every fused layer's backward (cProfile profiles the calling thread), so its table is the python time of
# the backward functions themselves (engine overhead between nodes excluded)
from zs3_amd import functional as Fz
prb = cProfile.Profile()
orig = Fz._ConvBnAct.backward
import time
spent = [0.0, 0]
def wrapped(ctx, *a):
    t0 = time.perf_counter()
    prb.enable()
    try:
        return orig(ctx, *a)
    finally:
        prb.disable()
        spent[0] += time.perf_counter() - t0; spent[1] += 1
Fz._ConvBnAct.backward = staticmethod(wrapped)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"\n==== backward of the fused layers: {spent[1] // steps} nodes per step, {1e3 * spent[0] / steps:.2f} ms per step inside them (profiled); "
      f"whole step host {1e3 * (t1 - t0) / steps:.2f} ms")
pstats.Stats(prb).sort_stats("tottime").print_stats(30)
