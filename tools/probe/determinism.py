"""Is the training step bit-reproducible?  Runs forward + backward of the same model on the same batch several times and compares
logits and every parameter gradient bitwise with the first run; reports the first tensors (in module order) that differ.
ZS3_STORAGE=bf16 selects the 2-byte mode.  usage: determinism.py [repeats] [B] [size]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.utils.loss import SegmentationLosses
from zs3_amd.utils.synthetic import make_batch

if os.environ.get("ZS3_STORAGE") == "bf16":
    ops.set_storage(torch.bfloat16)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else 513
dev = torch.device("cuda:0")
torch.manual_seed(1)
m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = float(os.environ.get("DROP", mod.p))
crit = SegmentationLosses(cuda=True).build_loss("ce")
b = make_batch(B, S, 21, [10, 14], seed=1, device=dev)
acts = {}
if os.environ.get("HOOKS"):
    def mk(name):
        def hook(mod, inp, out):
            if torch.is_tensor(out):
                acts.setdefault(name, []).append(out.detach().float().clone())
        return hook
    for name, mod in m.named_modules():
        if name and name.count(".") <= int(os.environ.get("HOOKS")):
            mod.register_forward_hook(mk(name))


def run():
    from zs3_amd import functional as Fz
    Fz._rng = None if hasattr(Fz, "_rng") else None      # same dropout seed stream in every repeat
    torch.manual_seed(7)
    for p in m.parameters():
        p.grad = None
    out = m(b["image"])
    loss = crit(out, b["label"])
    loss.backward()
    torch.cuda.synchronize()
    return out.detach().clone(), loss.item(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


ref = run()
ref_acts = {k: v[-1] for k, v in acts.items()}
for r in range(1, reps):
    acts.clear()
    out, loss, grads = run()
    bad_a = [k for k, v in acts.items() if not torch.equal(v[-1], ref_acts[k])]
    bad = [n for n in grads if not torch.equal(grads[n], ref[2][n])]
    print(f"repeat {r}: loss {loss!r} (first {ref[1]!r}); logits equal: {torch.equal(out, ref[0])}; "
          f"{len(bad)} of {len(grads)} gradients differ" + (f"; activations that differ first: {bad_a[:6]}" if acts else ""))
    if bad:
        worst = max(bad, key=lambda n: ((grads[n] - ref[2][n]).abs().max() / ref[2][n].abs().max().clamp_min(1e-30)).item())
        print("   last in module order that differ:", bad[-4:], " first:", bad[:4])
        print("   worst:", worst, ((grads[worst] - ref[2][worst]).abs().max() / ref[2][worst].abs().max()).item())
