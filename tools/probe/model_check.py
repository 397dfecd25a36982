import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, torch.nn as nn, numpy as np
import zs3_oracle as zo
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.utils.loss import SegmentationLosses, GMMNLoss
dev = torch.device("cuda:0")
def rel(a, b): return ((a.double().cpu()-b.double().cpu()).abs().max() / b.double().abs().max().cpu()).item()
torch.manual_seed(1)
m = DeepLab(num_classes=21, pretrained=False, sync_bn=False)
for name, mod in m.named_modules():
    if name.endswith("bn3"): mod.weight.data.fill_(0.1)
ref = zo.DeepLab(num_classes=21, pretrained=False); ref.load_state_dict(m.state_dict())
m = m.to(dev)
print("weights channels_last after .to:", m.backbone.layer1[0].conv2.weight.is_contiguous(memory_format=torch.channels_last))
b = zo.make_synthetic_batch(2, 65, seed=7, with_label_emb=False)
x, y = b["image"], b["label"]
m.eval(); ref.eval()
with torch.no_grad():
    t=time.time(); out = m(x.to(dev)); torch.cuda.synchronize(); print("eval fwd time", time.time()-t)
    r = ref(x)
print("eval logits rel", rel(out, r), "argmax mismatch", (out.argmax(1).cpu() != r.argmax(1)).sum().item(), "of", r.argmax(1).numel())
g = np.load(os.path.join(ROOT, "tests/golden/deeplab_forward.npz"))

# train mode, no dropout
for mod in list(m.modules()) + list(ref.modules()):
    if isinstance(mod, nn.Dropout): mod.p = 0.0
m.train(); ref.train()
w = torch.ones(21); w[[10,14]] = 100.0
out = m(x.to(dev)); loss = SegmentationLosses(weight=w.to(dev), cuda=True).build_loss("ce")(out, y.to(dev))
r = ref(x); rl = zo.SegmentationLosses(weight=w).build_loss("ce")(r, y)
print("train logits rel", rel(out.detach(), r.detach()), "loss", loss.item(), rl.item())
loss.backward(); rl.backward()
torch.cuda.synchronize()
worst = []
for (k, p), (k2, p2) in zip(m.named_parameters(), ref.named_parameters()):
    assert k == k2
    if p.grad is None: worst.append((float('inf'), k)); continue
    worst.append((rel(p.grad, p2.grad), k))
worst.sort(reverse=True)
print("worst grad rel:", worst[:8])
print("median grad rel:", worst[len(worst)//2])
for k in ["backbone.bn1.running_mean", "aspp.bn1.running_var", "decoder.last_conv.5.running_mean", "backbone.layer3.5.bn2.running_var"]:
    print(k, rel(m.state_dict()[k], ref.state_dict()[k]))
print("nbt", m.state_dict()["backbone.bn1.num_batches_tracked"].item())
