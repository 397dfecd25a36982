"""Which torch copies does one GMMN training step (B=16, 513x513, the bench.py configuration) launch, from which line, how large?
Attribution of the `direct_copy` / copyBuffer rows of profiles/r*_gmmn_kernel_stats.md.  Timed per call with events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, collections, traceback
from torch.utils._python_dispatch import TorchDispatchMode
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.modeling.gmmn import GMMNnetwork
from zs3_amd.gmmn_trainer import GMMNStep
from zs3_amd.optim import SGD, Adam
from zs3_amd.utils.loss import SegmentationLosses
from zs3_amd.utils.synthetic import make_batch

dev = torch.device("cuda:0")
B, S = int(os.environ.get("B", 16)), int(os.environ.get("S", 513))
unseen = [10, 14]
seen = [c for c in range(21) if c not in unseen]
torch.manual_seed(1)
model = DeepLab(num_classes=21, pretrained=False).to(dev).train()
opt = SGD([{"params": model.get_1x_lr_params(), "lr": 0.007}, {"params": model.get_10x_lr_params(), "lr": 0.07}], momentum=0.9, weight_decay=5e-4)
gen = GMMNnetwork(300, 300, 256, 256).to(dev).train()
opt_g = Adam(gen.parameters(), lr=2e-4)
w = torch.ones(21, device=dev); w[unseen] = 100.0
crit = SegmentationLosses(weight=w, cuda=True).build_loss("ce")
log = []


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        hit = any(k in name for k in ("copy_", "clone", "contiguous", "_to_copy", "index", "permute_copy"))
        if hit:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        out = func(*args, **(kwargs or {}))
        if hit:
            e1.record()
            st = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()
                  if "zs3_amd" in f.filename or "probe" in f.filename][-2:]
            a0 = args[0] if args and hasattr(args[0], "shape") else None
            log.append((name, tuple(st), tuple(a0.shape) if a0 is not None else None, str(a0.dtype) if a0 is not None else "",
                        str(a0.device.type) if a0 is not None else "", e0, e1))
        return out


with Spy():
    gb = make_batch(B, S, 21, unseen, seed=101, with_label_emb=True, device=dev)
torch.cuda.synchronize()
print("== setup (make_batch with the reference-faithful label_emb [B,300,S,S])")
for name, st, shp, dt, dv, e0, e1 in log:
    if e0.elapsed_time(e1) > 0.05:
        print(f"  {e0.elapsed_time(e1):8.3f} ms  {name}  {shp} {dt} {dv}  {st}")
log.clear()
stepper = GMMNStep(model, gen, opt, opt_g, crit, seen=seen, unseen=unseen, noise="device")
fn = lambda: stepper(gb["image"], gb["label"], gb["label_emb"], next_image=gb["image"])
for _ in range(3):
    fn()
torch.cuda.synchronize()
with Spy():
    fn()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, st, shp, dt, dv, e0, e1 in log:
    k = (name, st, shp, dt, dv)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
print("== one steady-state step: torch copy-like ops (count, total ms between events, op, shape, where)")
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {n:4d} x  {ms:8.3f} ms  {k[0]}  {k[2]} {k[3]} {k[4]}  {k[1]}")
