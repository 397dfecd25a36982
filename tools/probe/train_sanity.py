"""40 supervised steps on one fixed synthetic batch (B=8, 257x257): the loss must fall monotonically-ish and stay finite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.optim import SGD
from zs3_amd.utils.loss import SegmentationLosses
from zs3_amd.utils.synthetic import make_batch
from zs3_amd.utils.metrics import Evaluator
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
opt = SGD([{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}], momentum=0.9, weight_decay=5e-4)
crit = SegmentationLosses(cuda=True).build_loss("ce")
b = make_batch(8, 257, seed=5, device=dev)
losses = []
for i in range(40):
    opt.zero_grad(); out = m(b["image"]); loss = crit(out, b["label"]); loss.backward(); opt.step()
    losses.append(loss.item())
print("losses:", " ".join(f"{l:.3f}" for l in losses[::4]), "last", f"{losses[-1]:.4f}")
assert all(l == l and l < 50 for l in losses) and losses[-1] < 0.95 * losses[0], "loss does not fall"
m.eval(); ev = Evaluator(21)
with torch.no_grad():
    ev.add_batch_logits(b["label"], m(b["image"]))
print("train-batch pixel acc after 40 steps (eval mode):", ev.Pixel_Accuracy(), "mIoU", ev.Mean_Intersection_over_Union()[0])
