"""VERDICT r3 #3c, second half: what would the default-init train step (tests/test_gpu_model.py: _default_init_train_step) deliver
with an fp16 hi/lo operand split instead of bf16 hi/lo?  CPU emulation: every nn.Conv2d of the fp32 ORACLE network is replaced by an
autograd function whose forward / data-gradient / weight-gradient are three fp32-accumulated convolutions of split operands
(lo*hi + hi*lo + hi*hi, the kernels' product order).  Errors are the test's: max-abs relative against the fp64 oracle, in units of
the reference's own fp32 error (the golden file).  Runs in the build container (no GPU): python tools/probe/split_emulation.py"""
import copy, os, sys
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import zs3_oracle as zo
from zs3_amd.modeling.deeplab import DeepLab

MODE = {"fwd": None, "bwd": None}      # None = plain fp32; (dtype, scaled)


def split(x, dt, scaled):
    s = 1.0
    if scaled:
        amax = x.abs().max()
        if amax > 0:
            s = 2.0 ** (8 - torch.floor(torch.log2(amax)).item())
    xs = x * s
    hi = xs.to(dt).float()
    lo = (xs - hi).to(dt).float()
    return hi, lo, s


def prod3(fn, a, b, cfg):
    """cfg: None (plain fp32) or (dtype, scaled): scaled False = operands as they are, True = a power-of-two scale per tensor on
    both, a float = that fixed scale on the SECOND operand only (forward: the weights -- what the product's fp16 plane does)"""
    if cfg is None:
        return fn(a, b)
    dt, scaled = cfg
    if isinstance(scaled, float):
        ah, al, _ = split(a, dt, False)
        bh, bl, _ = split(b * scaled, dt, False)
        return (fn(al, bh) + fn(ah, bl) + fn(ah, bh)) / scaled
    ah, al, sa = split(a, dt, scaled)
    bh, bl, sb = split(b, dt, scaled)
    return (fn(al, bh) + fn(ah, bl) + fn(ah, bh)) / (sa * sb)


class EmuConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding, dilation):
        ctx.save_for_backward(x, w)
        ctx.geo = (stride, padding, dilation)
        return prod3(lambda a, b: F.conv2d(a, b, None, stride, padding, dilation), x, w, MODE["fwd"])

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, dilation = ctx.geo
        dx = prod3(lambda a, b: torch.nn.grad.conv2d_input(x.shape, b, a, stride, padding, dilation), dy, w, MODE["bwd"])
        dw = prod3(lambda a, b: torch.nn.grad.conv2d_weight(b, w.shape, a, stride, padding, dilation), dy, x, MODE["bwd"])
        return dx, dw, None, None, None


def emu_forward(self, x, w, b):
    y = EmuConv.apply(x, w, self.stride, self.padding, self.dilation)
    return y if b is None else y + b.view(1, -1, 1, 1)


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def evaluate(modes):
    """[(name, forward split, backward split)] -> (reference's own errors, [(name, errors vs fp64, logits vs goldens)])"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "deeplab_forward.npz"))
    torch.manual_seed(1)
    m = DeepLab(num_classes=21, pretrained=False)
    ref = zo.DeepLab(num_classes=21, pretrained=False)
    ref.load_state_dict(m.state_dict())
    for mod in ref.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    b = zo.make_synthetic_batch(2, 65, seed=7, with_label_emb=False)
    w = torch.ones(21)
    w[[10, 14]] = 100.0
    ref64 = copy.deepcopy(ref).double().train()
    r64 = ref64(b["image"].double())
    zo.SegmentationLosses(weight=w.double()).build_loss("ce")(r64, b["label"]).backward()
    gold, gp, gs = torch.from_numpy(g["train_logits"]), torch.from_numpy(g["grad_pred_w"]), torch.from_numpy(g["grad_stem_w"])
    e_ref = (rel(gold, r64), rel(gp, ref64.decoder.pred_conv.weight.grad), rel(gs, ref64.backbone.conv1.weight.grad[:8]))
    plain = nn.Conv2d._conv_forward
    results = []
    for name, fwd, bwd in modes:
        MODE["fwd"], MODE["bwd"] = fwd, bwd
        net = copy.deepcopy(ref).train()
        nn.Conv2d._conv_forward = emu_forward
        try:
            out = net(b["image"])
            zo.SegmentationLosses(weight=w).build_loss("ce")(out, b["label"]).backward()
        finally:
            nn.Conv2d._conv_forward = plain
        e = (rel(out, r64), rel(net.decoder.pred_conv.weight.grad, ref64.decoder.pred_conv.weight.grad),
             rel(net.backbone.conv1.weight.grad[:8], ref64.backbone.conv1.weight.grad[:8]))
        results.append((name, e, rel(out, gold)))
    return e_ref, results


BF, FH = torch.bfloat16, torch.float16
MODES = [("plain fp32 convolutions (this CPU's summation order)", None, None),
         ("bf16 hi/lo x3, forward and backward (the product's bf16x3)", (BF, False), (BF, False)),
         ("fp16 hi/lo x3 forward (unscaled), bf16 hi/lo x3 backward", (FH, False), (BF, False)),
         ("fp16 hi/lo x3 forward with the weights carried as 2^6 w, bf16 hi/lo x3 backward (the product since round 4)", (FH, 64.0), (BF, False)),
         ("fp16 hi/lo x3 forward (unscaled) and backward (power-of-two scale per tensor)", (FH, False), (FH, True)),
         ("fp16 hi/lo x3 forward and backward, both with a power-of-two scale per tensor", (FH, True), (FH, True))]


def main():
    e_ref, results = evaluate(MODES)
    print(f"reference fp32 (goldens) vs fp64: logits {e_ref[0]:.2e}, classifier gradient {e_ref[1]:.2e}, stem gradient {e_ref[2]:.2e}")
    print("| arithmetic of the 114 convolutions | logits vs fp64 (x reference) | classifier gradient (x reference) | stem gradient (x reference) | logits vs goldens |")
    print("|---|---|---|---|---|")
    for name, e, vs_gold in results:
        print(f"| {name} | {e[0]:.2e} ({e[0] / e_ref[0]:.1f}x) | {e[1]:.2e} ({e[1] / e_ref[1]:.1f}x) | {e[2]:.2e} ({e[2] / e_ref[2]:.1f}x) | {vs_gold:.2e} |")


if __name__ == "__main__":
    main()
