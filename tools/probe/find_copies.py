import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, collections, traceback
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.optim import SGD
from zs3_amd.utils.loss import SegmentationLosses
from zs3_amd.utils.synthetic import make_batch
dev = torch.device("cuda:0")
m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
opt = SGD([{"params": m.get_1x_lr_params(), "lr": 1e-3}, {"params": m.get_10x_lr_params(), "lr": 1e-2}], momentum=0.9, weight_decay=5e-4)
crit = SegmentationLosses(cuda=True).build_loss("ce")
b = make_batch(2, 129, seed=3, device=dev)
def step():
    opt.zero_grad(); loss = crit(m(b["image"]), b["label"]); loss.backward(); opt.step()
step(); step()
cnt = collections.Counter()
from torch.utils._python_dispatch import TorchDispatchMode
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy_", "clone", "contiguous", "_to_copy", "zero_", "fill_", "add")):
            st = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack() if "zs3_amd" in f.filename or "probe" in f.filename][-3:]
            shp = tuple(args[0].shape) if args and hasattr(args[0], "shape") else None
            cnt[(name, tuple(st), shp if len(str(shp)) < 30 else None)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    step()
torch.cuda.synchronize()
for k, v in cnt.most_common(40):
    print(v, k)
