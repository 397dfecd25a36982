import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.utils.loss import SegmentationLosses
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 513
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.manual_seed(1)
m = DeepLab(num_classes=21, pretrained=False, sync_bn=False).to(dev).train()
params = [{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}]
opt = torch.optim.SGD(params, momentum=0.9, weight_decay=5e-4)
crit = SegmentationLosses(cuda=True).build_loss("ce")
x = torch.randn(B, 3, S, S, device=dev); y = torch.randint(0, 21, (B, S, S), device=dev).float()
def step():
    opt.zero_grad(); out = m(x); loss = crit(out, y); loss.backward(); opt.step(); return loss
for _ in range(2): step()
torch.cuda.synchronize()
t = time.time()
for _ in range(steps): l = step()
torch.cuda.synchronize(); dt = (time.time()-t)/steps
print(f"B={B} S={S}: {dt*1e3:.1f} ms/step, {B/dt:.1f} img/s, loss {l.item():.4f}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
# split fwd / bwd
torch.cuda.synchronize(); t=time.time()
for _ in range(steps):
    with torch.no_grad(): out = m(x)
torch.cuda.synchronize(); print(f"fwd-only (no_grad, train BN): {(time.time()-t)/steps*1e3:.1f} ms")
t=time.time()
for _ in range(steps):
    opt.zero_grad(); out = m(x); loss = crit(out, y)
torch.cuda.synchronize(); tf=(time.time()-t)/steps
print(f"fwd+loss with graph: {tf*1e3:.1f} ms")
