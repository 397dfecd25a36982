"""How much of the step is launch overhead?  eager fwd+bwd vs the same captured in one hipGraph (B=16, 513^2).
Everything (eager warm-up included) runs on a non-default stream: once autograd has bound a parameter's AccumulateGrad
node to the legacy null stream, a later capture dies in hipStreamEndCapture."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, faulthandler
faulthandler.enable()
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.utils.loss import SegmentationLosses
from zs3_amd.utils.synthetic import make_batch
import zs3_amd.functional as Fz
from zs3_amd import ops
from zs3_amd.optim import SGD
if os.environ.get("ZS3_STORAGE") == "bf16":
    ops.set_storage(torch.bfloat16)
dev = torch.device("cuda:0")
B, S = int(os.environ.get("PB", 16)), int(os.environ.get("PS", 513))
Fz.WGRAD_SIDE_STREAM = os.environ.get("SIDE", "1") == "1"
torch.manual_seed(1)
work_stream = torch.cuda.Stream()
with torch.cuda.stream(work_stream):
    m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    b = make_batch(B, S, seed=3, device=dev)
    img, lab = b["image"], b["label"]
    opt = SGD([{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}], momentum=0.9, weight_decay=5e-4)
    def fb():
        loss = crit(m(img), lab)
        loss.backward()
        if os.environ.get("OPT", "1") == "1":
            opt.step()
        return loss
    def timeit(fn, n=5):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    for p in m.parameters(): p.grad = None
    fb(); fb()
    print("eager fwd+bwd ms", timeit(fb), flush=True)
    for p in m.parameters(): p.grad = None
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = fb()
    g.replay(); torch.cuda.synchronize()
    print("graph fwd+bwd ms", timeit(g.replay), "loss", loss.item(), flush=True)
    print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
