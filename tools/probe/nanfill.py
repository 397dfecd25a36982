"""Find reads of never-written device memory: every torch.empty / empty_like / empty_strided allocation is pre-filled with NaN
(floating types) or 0xFF bytes, then one forward + backward runs and the first product-level op whose OUTPUT holds a NaN is reported
with its arguments.  ZS3_STORAGE=bf16 selects the 2-byte mode.  usage: nanfill.py [B] [size]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops, functional as Fz
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.utils.loss import SegmentationLosses
from zs3_amd.utils.synthetic import make_batch

if os.environ.get("ZS3_STORAGE") == "bf16":
    ops.set_storage(torch.bfloat16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 513
dev = torch.device("cuda:0")
torch.manual_seed(1)
m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
crit = SegmentationLosses(cuda=True).build_loss("ce")
b = make_batch(B, S, 21, [10, 14], seed=1, device=dev)

_empty, _empty_like, _empty_strided = torch.empty, torch.empty_like, torch.empty_strided
def poison(t):
    if t.is_cuda:
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(255)
    return t
torch.empty = lambda *a, **k: poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: poison(_empty_like(*a, **k))
torch.empty_strided = lambda *a, **k: poison(_empty_strided(*a, **k))

found = []
def has_nan(t):
    return torch.is_tensor(t) and t.is_cuda and t.dtype.is_floating_point and bool(torch.isnan(t.float()).any())
def describe(x):
    if torch.is_tensor(x):
        return f"T{tuple(x.shape)}:{str(x.dtype).replace('torch.', '')}:st{tuple(x.stride())}{':NaN' if has_nan(x) else ''}"
    if isinstance(x, (list, tuple)):
        return "[" + ", ".join(describe(v) for v in x) + "]"
    return repr(x) if not hasattr(x, "f_pk") else "planes"
def wrap(mod, name):
    fn = getattr(mod, name)
    def w(*a, **k):
        out = fn(*a, **k)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        if len(found) < 6 and any(has_nan(o) for o in outs):
            ins_nan = any(has_nan(v) for v in list(a) + list(k.values()))
            found.append(f"{name}: output has NaN (inputs {'ALSO have' if ins_nan else 'have no'} NaN)\n      args {describe(a)}\n      kw {{{', '.join(f'{kk}: {describe(v)}' for kk, v in k.items())}}}\n      out {describe(outs)}")
        return out
    setattr(mod, name, w)
for name in ("conv_igemm", "conv2d_wgrad", "affine_act", "bn_fwd_finalize", "bn_bwd_finalize", "bn_act_bwd", "bn_bwd_stats", "maxpool_fwd",
             "maxpool_bwd", "bilinear_fwd", "bilinear_bwd", "sum_n", "group_colsum", "colsum", "dropout", "cast"):
    if hasattr(ops, name):
        wrap(ops, name)
out = m(b["image"])
loss = crit(out, b["label"])
print("forward done: loss", loss.item(), "| findings so far:", len(found))
loss.backward()
torch.cuda.synchronize()
bad = [n for n, p in m.named_parameters() if p.grad is not None and has_nan(p.grad)]
print("gradients with NaN:", len(bad), bad[:8])
for f in found:
    print(" *", f)
