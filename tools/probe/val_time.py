"""N4: cost of one validation batch's metric update (B=16, 513x513, 21 classes) -- reference flow vs device flow."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from zs3_amd import functional as Fz, ops
from zs3_amd.utils.metrics import Evaluator
dev = torch.device("cuda:0")
lo = torch.randn(16, 21, 129, 129, device=dev).contiguous(memory_format=torch.channels_last)
gt = torch.randint(0, 21, (16, 513, 513), device=dev).float()
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ev = Evaluator(21)
def ref_flow():
    out = ops.nchw(Fz.bilinear(ops.nhwc(lo), (513, 513)))
    pred = np.argmax(out.cpu().numpy(), axis=1)                       # train_pascal.py:130-132
    ev.add_batch(gt.cpu().numpy(), pred)
def dev_full():
    out = ops.nchw(Fz.bilinear(ops.nhwc(lo), (513, 513)))
    ev.add_batch_logits(gt, out)
def dev_fused():
    ev.add_batch_logits(gt, lo)
print(f"reference flow (upsample on GPU, D2H 354 MB, numpy argmax + bincount): {t(ref_flow, 2):.1f} ms")
print(f"device, full-resolution logits (upsample kernel + fused argmax/histogram): {t(dev_full):.3f} ms")
print(f"device, low-resolution logits (upsample fused too):                      {t(dev_fused):.3f} ms")
