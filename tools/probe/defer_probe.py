"""Which layers hand their BatchNorm-apply to their consumer (functional.DEFER_BN_APPLY), and how many BN-apply launches a step has."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from zs3_amd import ops, functional as Fz
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.utils.loss import SegmentationLosses
from zs3_amd.optim import SGD
dev = torch.device("cuda:0")
torch.manual_seed(1)
m = DeepLab(num_classes=21, pretrained=False, sync_bn=False).to(dev).train()
opt = SGD([{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}], momentum=0.9, weight_decay=5e-4)
crit = SegmentationLosses(cuda=True).build_loss("ce")
x = torch.randn(16, 3, 513, 513, device=dev); y = torch.randint(0, 21, (16, 513, 513), device=dev).float()
n = [0]
real = ops.affine_act
def counted(*a, **k):
    n[0] += 1
    return real(*a, **k)
ops.affine_act = counted
def step():
    opt.zero_grad(); out = m(x); loss = crit(out, y); loss.backward(); opt.step(); return loss
step(); n[0] = 0; l = step(); torch.cuda.synchronize()
print("affine_act launches per step:", n[0], " loss", l.item())
yes = [k for k, v in Fz._defer_choice.items() if v]; no = [k for k, v in Fz._defer_choice.items() if not v]
print("deferred layer geometries:", len(yes), " not deferred:", len(no))
for k in yes: print("  defer", k[0], k[1])
for k in no: print("  keep ", k[0], k[1])
for _ in range(3): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print(f"{(time.perf_counter() - t) * 100:.2f} ms/step")
