"""Whole-model parity against the CPU oracle at the sizes that matter (VERDICT r5 "next" #3):

* 513 x 513 -- the benchmark's own resolution (BASELINE configs[1]; B = 2 so that the fp64 oracle finishes in seconds): the decoder's
  W = 129 strips, the 129 -> 513 resize in front of the CE, the stem's BatchNorm finalize over 2 x 257 x 257 positions and the tile
  rules at M = 33 282 ... 526 338 are reached through the ASSEMBLED network here, not only through per-layer checks;
* 312 x 312 -- the reference's default training crop (train_pascal.py:203-204): the chain 312 -> 156 -> 78 -> 39 -> 20 -> 20 has even
  sizes, the decoder's resize is 20 -> 78 (scale 3.9, not 4) and the final one 78 -> 312;
* output_stride = 8 (resnet.py:72-74: strides [1, 2, 1, 1], dilations [1, 1, 2, 4], multi-grid 4 x [1, 2, 4]; aspp.py:49-50: rates
  [1, 12, 24, 36]) -- the product ships `_STAGES[8]`, this is the test that runs it.

The judge of every comparison is the fp64 evaluation of the oracle (`ref.double()`); the fp32 oracle's own distance from it is
printed next to ours.  Tolerances are 3x what the kernels delivered on MI355X when the test was written (the delivered value is in
the comment next to each bound, and every run prints the current ones)."""
import copy
import os

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))     # the oracle's convolutions: more threads than this run slower
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def build_pair(num_classes=21, tame=True, **kw):
    import zs3_oracle as zo
    from zs3_amd.modeling.deeplab import DeepLab
    torch.manual_seed(1)
    m = DeepLab(num_classes=num_classes, pretrained=False, sync_bn=False, **kw)
    if tame:    # trained-ResNet-like residual gains: a default-init train-mode network is chaotic (tests/test_gpu_model.py header)
        for name, mod in m.named_modules():
            if name.endswith("bn3"):
                mod.weight.data.fill_(0.1)
    ref = zo.DeepLab(num_classes=num_classes, pretrained=False, **kw)
    ref.load_state_dict(m.state_dict())
    for mod in list(m.modules()) + list(ref.modules()):
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    return m, ref


def _argmax_agrees(out, gold, tol):
    """argmax identical wherever the fp64 oracle's own top-2 margin exceeds twice the logit tolerance -> fraction of such pixels"""
    gold = gold.detach().double().cpu()
    top2 = gold.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * tol * gold.abs().max()
    assert torch.equal(out.argmax(1).cpu()[safe], gold.argmax(1)[safe])
    return safe.double().mean().item()


# (size, delivered logits error vs fp64, bound) -- eval mode, default-initialised weights (running statistics 0 / 1)
EVAL_CASES = {513: 6e-5, 312: 6e-5}     # delivered on MI355X: 1.93e-5 / 1.96e-5 (the oracle's own fp32: 1.2e-6 / 1.5e-6)


@pytest.mark.parametrize("size", sorted(EVAL_CASES))
def test_eval_logits_and_argmax_at_the_real_sizes(dev, golden, size):
    """B = 2 eval forward of the assembled DeepLabv3+ against the oracle in fp64 (and the oracle's own fp32 next to it), and
    against the REFERENCE's own output at this size (tests/golden/sizes.npz: subsampled logits, full argmax map)."""
    import numpy as np
    import zs3_oracle as zo
    g = golden("sizes.npz")
    m, ref = build_pair(tame=False)
    b = zo.make_synthetic_batch(2, size, seed=size, with_label_emb=False)
    m = m.to(dev).eval()
    ref.eval()
    with torch.no_grad():
        out = m(b["image"].to(dev))
        r32 = ref(b["image"])
        r64 = copy.deepcopy(ref).double()(b["image"].double())
    assert out.shape == r64.shape == (2, 21, size, size)
    e, e32 = rel(out, r64), rel(r32, r64)
    frac = _argmax_agrees(out, r64, EVAL_CASES[size])
    print(f"\n[parity {size}x{size} eval] logits vs fp64: ours {e:.2e}, the oracle's own fp32 {e32:.2e}; argmax identical on the "
          f"{100 * frac:.2f} % of pixels outside the 2 x {EVAL_CASES[size]:.0e} margin")
    assert e < 1e-3                      # north-star tolerance
    assert e < EVAL_CASES[size]          # 3x delivered
    assert frac > 0.99
    gold = torch.from_numpy(g[f"eval{size}_logits_sub"])
    eg = rel(out[:, :, ::8, ::8], gold)
    am, gam = out.argmax(1).cpu().numpy(), g[f"eval{size}_argmax"].astype(np.int64)
    print(f"[parity {size}x{size} eval] vs the reference's golden logits: {eg:.2e}; argmax differs on {int((am != gam).sum())} of "
          f"{am.size} pixels")
    assert eg < EVAL_CASES[size]
    assert (am != gam).mean() < 1e-3     # (the reference's own fp32 argmax: near-ties may fall either way; the fp64 check above is exact)


# train mode, tamed residual gains.  Bounds = 3x delivered (comment: delivered when written)
TRAIN_TOL = {
    # delivered at 513: logits 1.98e-4 (the oracle's own fp32: 1.0e-5), loss 1.9e-6, worst running statistic 2.1e-4
    513: {"logits": 6e-4, "loss": 1e-5, "running": 6.5e-4,
          "grads": {"decoder.pred_conv.weight": 2.5e-4,          # 7.7e-5  (oracle fp32: 3.6e-6)
                    "decoder.pred_conv.bias": 5e-5,              # 1.4e-5  (1.8e-6)
                    "decoder.last_conv.0.weight": 1.2e-2,        # 3.8e-3  (9.4e-4)
                    "aspp.conv1.weight": 2.7e-2,                 # 8.9e-3  (9.5e-4)
                    "backbone.layer4.2.conv3.weight": 2.4e-2,    # 7.9e-3  (1.6e-3)
                    "backbone.layer1.0.conv1.weight": 2.6e-2,    # 8.6e-3  (2.9e-3)
                    "backbone.conv1.weight": 2.6e-2,             # 8.4e-3  (2.9e-3)
                    "backbone.bn1.weight": 3e-2}},               # 9.9e-3  (3.5e-3)
    # delivered at 312: logits 1.38e-4 (7.5e-6), loss 1.5e-6, worst running statistic 2.1e-4
    312: {"logits": 4.2e-4, "loss": 1e-5, "running": 6.5e-4,
          "grads": {"decoder.pred_conv.weight": 2.8e-4,          # 9.3e-5  (4.4e-6)
                    "decoder.pred_conv.bias": 8e-5,              # 2.7e-5  (1.1e-6)
                    "decoder.last_conv.0.weight": 1.2e-2,        # 3.9e-3  (7.4e-4)
                    "aspp.conv1.weight": 2.8e-2,                 # 9.2e-3  (6.6e-4)
                    "backbone.layer4.2.conv3.weight": 2.6e-2,    # 8.4e-3  (1.5e-3)
                    "backbone.layer1.0.conv1.weight": 2.7e-2,    # 8.9e-3  (3.6e-3)
                    "backbone.conv1.weight": 2.7e-2,             # 9.0e-3  (3.1e-3)
                    "backbone.bn1.weight": 2.6e-2}},             # 8.5e-3  (2.7e-3)
}
# (The gradient errors below the decoder are ~3x the fp32 oracle's own distance from fp64: the backward pass multiplies bf16x3
# products (2^-16-class) and every ReLU whose pre-activation sits within that of zero may take the other
# branch; the classifier's gradient, which sees no ReLU below it in the backward chain, is at 8e-5.)


@pytest.mark.parametrize("size", sorted(TRAIN_TOL))
def test_train_forward_ce_and_gradients_at_the_real_sizes(dev, golden, size):
    """One train-mode forward + weighted CE + backward at B = 2 against the fp64 oracle: logits (batch statistics in all 113
    BatchNorm layers), loss, running statistics, and the gradients from the classifier down to the stem."""
    import zs3_oracle as zo
    from zs3_amd.utils.loss import SegmentationLosses
    tol = TRAIN_TOL[size]
    m, ref = build_pair(tame=True)
    ref64 = copy.deepcopy(ref).double()
    b = zo.make_synthetic_batch(2, size, seed=1000 + size, with_label_emb=False)
    x, y = b["image"], b["label"]
    w = torch.ones(21)
    w[[10, 14]] = 100.0
    m = m.to(dev).train()
    out = m(x.to(dev))
    loss = SegmentationLosses(weight=w.to(dev), cuda=True).build_loss("ce")(out, y.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    ref.train()
    r32 = ref(x)
    l32 = zo.SegmentationLosses(weight=w).build_loss("ce")(r32, y)
    l32.backward()
    g32 = {k: p.grad.double() for k, p in ref.named_parameters() if k in tol["grads"]}
    r32 = r32.detach()
    del ref
    ref64.train()
    r64 = ref64(x.double())
    l64 = zo.SegmentationLosses(weight=w.double()).build_loss("ce")(r64, y)
    l64.backward()
    e, e32 = rel(out, r64), rel(r32, r64)
    el = abs(loss.item() - l64.item()) / abs(l64.item())
    print(f"\n[parity {size}x{size} train] logits vs fp64: ours {e:.2e} (oracle fp32 {e32:.2e}); loss {loss.item():.6f} vs "
          f"{l64.item():.6f} (rel {el:.1e})")
    assert e < 1e-3 and e < tol["logits"]          # delivered: see the table in DESIGN.md section 5
    assert el < tol["loss"]
    sd, sd64 = m.state_dict(), ref64.state_dict()
    worst = max((rel(sd[k], sd64[k]), k) for k in sd if "running" in k)
    print(f"[parity {size}x{size} train] worst running statistic: {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < tol["running"], worst
    assert all(int(sd[k]) == 1 for k in sd if "num_batches_tracked" in k)
    params, p64 = dict(m.named_parameters()), dict(ref64.named_parameters())
    bad = []
    for k, bound in tol["grads"].items():
        g = p64[k].grad
        ours, own = rel2(params[k].grad, g), rel2(g32[k], g)
        print(f"[parity {size}x{size} train] d{k}: ours {ours:.2e}, oracle fp32 {own:.2e}, bound {bound:.0e}")
        if not ours < bound:
            bad.append((k, ours, bound))
    assert not bad, bad
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    if size == 312:     # the reference's own train-mode step at its default crop (tests/golden/sizes.npz)
        g = golden("sizes.npz")
        eg = rel(out[:, :, ::8, ::8], torch.from_numpy(g["train312_logits_sub"]))
        gp = rel2(params["decoder.pred_conv.weight"].grad, torch.from_numpy(g["train312_grad_pred_w"]))
        lg = abs(loss.item() - float(g["train312_loss"])) / abs(float(g["train312_loss"]))
        print(f"[parity 312x312 train] vs the reference's goldens: logits {eg:.2e}, loss {lg:.1e}, d pred_conv {gp:.2e}")
        assert eg < tol["logits"] and lg < tol["loss"] and gp < tol["grads"]["decoder.pred_conv.weight"]


def test_output_stride_8_eval_forward(dev, golden):
    """DeepLab(output_stride=8) (resnet.py:72-74, aspp.py:49-50) at 65 x 65: layer3 keeps 9 x 9 with dilation 2, layer4 runs
    dilations 4 / 8 / 16, ASPP rates 12 / 24 / 36 -- all beyond the 9 x 9 map's reach except through the centre tap, the dead-tap
    rule's extreme case.  Eval logits + argmax + the ASPP / backbone outputs against the fp64 oracle; and one train-mode step
    runs with finite gradients (train-mode logits against fp64 too: tamed gains)."""
    import zs3_oracle as zo
    from zs3_amd.utils.loss import SegmentationLosses
    m, ref = build_pair(tame=True, output_stride=8)
    b = zo.make_synthetic_batch(2, 65, seed=8, with_label_emb=False)
    x = b["image"]
    m = m.to(dev).eval()
    ref64 = copy.deepcopy(ref).double().eval()
    with torch.no_grad():
        out = m(x.to(dev))
        top, low = m.backbone(x.to(dev))
        r64 = ref64(x.double())
        rtop, rlow = ref64.backbone(x.double())
        r32 = ref.eval()(x)
    assert out.shape == r64.shape and top.shape == rtop.shape == (2, 2048, 9, 9) and low.shape == rlow.shape
    e, e32, et = rel(out, r64), rel(r32, r64), rel(top, rtop)
    frac = _argmax_agrees(out, r64, 6e-5)
    print(f"\n[parity OS8 65x65 eval] logits vs fp64: ours {e:.2e} (oracle fp32 {e32:.2e}), backbone output {et:.2e}, argmax "
          f"identical on {100 * frac:.2f} % of pixels")
    assert e < 6e-5 and et < 3e-5 and frac > 0.98          # delivered: 1.96e-5, 9.6e-6, 99.7 %
    g = golden("sizes.npz")
    eg = rel(out, torch.from_numpy(g["os8_logits"]))
    print(f"[parity OS8 65x65 eval] vs the reference's golden logits: {eg:.2e}")
    assert eg < 6e-5            # delivered: 1.91e-5
    m.train()
    ref64.train()
    out = m(x.to(dev))
    loss = SegmentationLosses(cuda=True).build_loss("ce")(out, b["label"].to(dev))
    loss.backward()
    r64 = ref64(x.double())
    l64 = zo.SegmentationLosses().build_loss("ce")(r64, b["label"])
    l64.backward()
    et = rel(out, r64)
    g = rel2(m.decoder.pred_conv.weight.grad, ref64.decoder.pred_conv.weight.grad)
    print(f"[parity OS8 65x65 train] logits {et:.2e}, loss {loss.item():.6f} vs {l64.item():.6f}, d pred_conv {g:.2e}")
    assert et < 2e-4 and abs(loss.item() - l64.item()) < 1e-5 * abs(l64.item()) and g < 1.1e-4     # delivered: 5.4e-5, <1e-6, 3.5e-5
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
