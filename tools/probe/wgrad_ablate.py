"""Diagnostic (not a product path): the supervised step with the weight-gradient launches removed, to see how much of the step
the wgrad side streams cost next to the main-stream chain.  usage: wgrad_ablate.py [steps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from zs3_amd import ops, functional as Fz
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.utils.loss import SegmentationLosses
if os.environ.get("ZS3_STORAGE") == "bf16":
    ops.set_storage(torch.bfloat16)
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(1)
m = DeepLab(num_classes=21, pretrained=False, sync_bn=False).to(dev).train()
params = [{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}]
from zs3_amd.optim import SGD
opt = SGD(params, momentum=0.9, weight_decay=5e-4)     # the fused update of the bench step (torch.optim.SGD: +1.5 ms GPU, 42 ms of host time)
crit = SegmentationLosses(cuda=True).build_loss("ce")
x = torch.randn(16, 3, 513, 513, device=dev); y = torch.randint(0, 21, (16, 513, 513), device=dev).float()
def step():
    opt.zero_grad(); out = m(x); loss = crit(out, y); loss.backward(); opt.step(); return loss
def run(tag):
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); print(f"{tag}: {(time.perf_counter() - t) / steps * 1e3:.2f} ms/step, loss {float(step()):.6f}", flush=True)
run("full step")
real = ops.conv2d_wgrad
cache = {}
def fake(dy, x, cout, cin, kh, kw, *a, out=None, **k):
    if out is not None: return out
    key = (cout, kh, kw, cin)
    if key not in cache: cache[key] = torch.zeros((cout, kh, kw, cin), device=dy.device)
    return cache[key]
ops.conv2d_wgrad = fake
run("no wgrad launches")
ops.conv2d_wgrad = real
def only_pw(dy, x, cout, cin, kh, kw, *a, out=None, **k):
    if kh == 1: return real(dy, x, cout, cin, kh, kw, *a, out=out, **k)
    return fake(dy, x, cout, cin, kh, kw, *a, out=out, **k)
ops.conv2d_wgrad = only_pw
run("only the 1x1 wgrads")
ops.conv2d_wgrad = real
# host time of a step: launches only, no synchronisation until the end of all steps
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps): step()
host = (time.perf_counter() - t) / steps
torch.cuda.synchronize(); print(f"host issue time per step (no sync): {host * 1e3:.2f} ms")
