import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, faulthandler
faulthandler.enable()
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.utils.loss import SegmentationLosses
from zs3_amd.utils.synthetic import make_batch
import zs3_amd.functional as Fz
Fz.WGRAD_SIDE_STREAM = False
dev = torch.device("cuda:0")
mode = sys.argv[1]
torch.manual_seed(1)
m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
crit = SegmentationLosses(cuda=True).build_loss("ce")
b = make_batch(2, 129, seed=3, device=dev)
img, lab = b["image"], b["label"]
def work():
    if mode == "fwd":
        return m(img).sum()
    if mode == "fwdloss":
        return crit(m(img), lab)
    if mode == "backbone":
        x, low = m.backbone(img)
        return x.sum() + low.sum()
    if mode == "backbone_bwd":
        x, low = m.backbone(img)
        l = x.sum() + low.sum(); l.backward(); return l
    if mode == "model_bwd":
        l = m(img).sum(); l.backward(); return l
    if mode == "aspp_bwd":
        with torch.no_grad():
            x, low = m.backbone(img)
        x = x.detach().requires_grad_(True)
        l = m.aspp(x).sum(); l.backward(); return l
    if mode == "dec_bwd":
        with torch.no_grad():
            x, low = m.backbone(img); x = m.aspp(x)
        x = x.detach().requires_grad_(True); low = low.detach().requires_grad_(True)
        l = m.decoder(x, low).sum(); l.backward(); return l
    if mode == "loss_bwd":
        with torch.no_grad():
            o = m(img)
        o = o.detach().requires_grad_(True)
        l = crit(o, lab); l.backward(); return l
    if mode == "full":
        l = crit(m(img), lab); l.backward(); return l
if os.environ.get("EAGER_FIRST") == "1":
    work(); work(); torch.cuda.synchronize()
    for p in m.parameters(): p.grad = None
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        for p in m.parameters(): p.grad = None
        work()
torch.cuda.current_stream().wait_stream(s)
for p in m.parameters(): p.grad = None
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = work()
g.replay(); torch.cuda.synchronize()
print(mode, "OK", float(out))
