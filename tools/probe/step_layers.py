"""In-step per-layer table of the conv launches (forward + data gradient) of the supervised training step at B=16, 513x513:
HIP events around EVERY conv launch of `steps` instrumented steps (ops.PROFILE + ops.PROFILE_GEOM), aggregated by
(kernel, geometry, epilogue).  The event pairs serialise kernel boundaries on the main stream (the step gets ~3 % longer); the
weight-gradient side streams keep running next to the launches, so the durations carry the step's contention.
usage: step_layers.py [steps] [batch] [classes] [bf16]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from zs3_amd import ops
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.optim import SGD
from zs3_amd.utils.loss import SegmentationLosses
from zs3_amd.utils.synthetic import make_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
C = int(sys.argv[3]) if len(sys.argv) > 3 else 21
if len(sys.argv) > 4 and sys.argv[4] == "bf16":
    ops.set_storage(torch.bfloat16)
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = DeepLab(num_classes=C, pretrained=False, sync_bn=False).to(dev).train()
opt = SGD([{"params": model.get_1x_lr_params(), "lr": 0.007}, {"params": model.get_10x_lr_params(), "lr": 0.07}], momentum=0.9,
          weight_decay=5e-4, nesterov=False)
crit = SegmentationLosses(cuda=True).build_loss("ce")
b = make_batch(B, 513, C, [10, 14], seed=1, device=dev)
def step():
    opt.zero_grad(); loss = crit(model(b["image"]), b["label"]); loss.backward(); opt.step(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
ops.PROFILE, ops.PROFILE_GEOM = [], []
for _ in range(steps): loss = step()
torch.cuda.synchronize()
prof, geom = ops.PROFILE, ops.PROFILE_GEOM
ops.PROFILE = ops.PROFILE_GEOM = None
agg = {}
for (tag, fl, e0, e1, cfg), g in zip(prof, geom):
    a = agg.setdefault((tag, cfg) + g, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] = fl
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values()) / steps
print(f"# conv launches inside the step, B={B}, {C} classes, loss {loss.item():.4f}; total {tot/1e3:.2f} ms per step over {len(prof)//steps} launches")
print("| launches/step | kernel | cfg | M | N | K | taps | s | d | dgrad | epilogue | io | avg us | TF | ms/step | algorithmic MB | GB/s |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
one = {1: 1, 2: 1}
for (tag, cfg, m, n, k, taps, s, d, dg, epi, io), (cnt, us, fl) in rows:
    eb = 2 if io & 2 else 4
    xb = 2 if io & 1 else 4
    mb = (m * s * s * k * xb + m * n * eb * (1 + ("res" in epi) + ("acc" in epi) + ("bnbwd" in epi)) + n * k * taps * 4) / 1e6
    print(f"| {cnt/steps:.0f} | {tag} | {cfg} | {m} | {n} | {k} | {taps} | {s} | {d} | {dg} | {epi} | {io} | {us/cnt:.1f} | {fl/(us/cnt)/1e6:.0f} | {us/steps/1e3:.3f} | {mb:.0f} | {mb/(us/cnt)*1e3:.0f} |")
