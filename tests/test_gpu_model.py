"""GPU parity of the assembled hot path (DeepLabv3+, GMMN, training steps) against the CPU oracle and the
golden vectors generated from the reference.  All compute goes through libzs3hip.so.

Conditioning note (DESIGN.md "Parity"): a randomly initialised ResNet-101 in train() mode on tiny inputs
is chaotic -- the reference itself moves by 1.7e-3 (logits) / up to 7e-2 (some gradients) between fp32 and
fp64.  Train-mode tests therefore set the residual-branch BN gains to 0.1 (what trained ResNets look like)
and judge gradients against the fp64 oracle relative to the fp32 oracle's own error."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


# Two SGD iterations against the fp64 oracle, relative L2 error of the accumulated update of representative weights.  Each
# tolerance is 3x what the kernels deliver on MI355X (delivered value in the comment; the test prints them), capped at the
# 3e-2 the test used before.  The error grows with the depth of the backward chain below the layer: every ReLU whose
# pre-activation sits within ~1e-5 of zero may take the other branch (tests/test_gpu_ops.py::test_stem_conv_at_full_size).
UPDATE_TOL = {
    "decoder.pred_conv.weight": 4e-4,           # 1.2e-4
    "decoder.pred_conv.bias": 1e-4,             # 2.4e-5
    "decoder.last_conv.4.weight": 9e-3,         # 3.0e-3
    "aspp.conv1.weight": 2e-2,                  # 6.7e-3
    "backbone.layer4.2.conv3.weight": 3e-2,     # 9.8e-3
    "backbone.layer1.0.conv1.weight": 3e-2,     # 1.6e-2 (cap)
    "backbone.conv1.weight": 3e-2,              # 1.7e-2 (cap)
    "backbone.bn1.weight": 3e-2,                # 1.6e-2 (cap)
}


# test_train_forward_backward_vs_oracle: bound on every parameter's gradient (relative L2 against the fp64 oracle), 3x the worst
# value delivered on MI355X (printed by the test: 3.03e-2 on backbone.layer3.5.bn3.bias = 24x the fp32 oracle's own 1.2e-3, median
# over the 320 parameters 1.2e-3; through round 5 the bound was max(300 x the fp32 oracle's own error, 2e-3), i.e. 0.37 there)
GRAD_TOL = 9e-2


def build_pair(num_classes=21, tame=True, dropout=0.0, **kw):
    import zs3_oracle as zo
    from zs3_amd.modeling.deeplab import DeepLab
    torch.manual_seed(1)
    m = DeepLab(num_classes=num_classes, pretrained=False, **kw)
    if tame:
        for name, mod in m.named_modules():
            if name.endswith("bn3"):
                mod.weight.data.fill_(0.1)
    ref = zo.DeepLab(num_classes=num_classes, pretrained=False, **{k: v for k, v in kw.items() if k != "sync_bn"})
    ref.load_state_dict(m.state_dict())
    for mod in list(m.modules()) + list(ref.modules()):
        if isinstance(mod, nn.Dropout):
            mod.p = dropout
    return m, ref


def test_eval_logits_and_argmax_vs_golden_and_oracle(dev, golden):
    """default-init model (seed 1), eval mode, 65x65: golden logits come from the reference itself"""
    import zs3_oracle as zo
    m, ref = build_pair(tame=False)
    g = golden("deeplab_forward.npz")
    b = zo.make_synthetic_batch(2, 65, seed=7, with_label_emb=False)
    m = m.to(dev).eval()
    with torch.no_grad():
        out = m(b["image"].to(dev))
        feat = m.forward_before_class_prediction(b["image"].to(dev))
    gold = torch.from_numpy(g["eval_logits"])
    assert out.shape == gold.shape
    assert rel(out, gold) < 1e-3           # north-star tolerance: logits within 1e-3 rel of the reference
    assert rel(out, gold) < 2e-4           # what bf16x3 actually delivers
    # argmax bit-exact wherever the reference's own top-2 margin exceeds the logit tolerance
    top2 = gold.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    safe = margin > 2e-4 * gold.abs().max()
    am = out.argmax(1).cpu()
    assert torch.equal(am[safe], torch.from_numpy(g["eval_argmax"].astype(np.int64))[safe])
    assert safe.float().mean() > 0.99
    assert np.allclose(feat.cpu()[:, :8, ::4, ::4].numpy(), g["eval_feat_slice"], rtol=2e-3, atol=2e-4 * float(np.abs(g["eval_feat_slice"]).max()))


def test_baseline_config0_forward_and_loss(dev, golden):
    """BASELINE.json configs[0]: forward + CE loss on one random 3x129x129 tensor against the reference's own output
    (tests/golden/config0_129.npz): logits within 1e-3 rel (delivered: < 2e-4), argmax identical on all 16641 pixels (the
    reference's smallest top-2 margin is 1.05), loss within 1e-3; train() mode with B = 1 raises like aspp.py:87."""
    import zs3_oracle as zo
    from zs3_amd.utils.loss import SegmentationLosses
    g = golden("config0_129.npz")
    m, _ = build_pair(tame=False)
    m = m.to(dev).eval()
    b = zo.make_synthetic_batch(1, 129, seed=129, with_label_emb=False)
    with torch.no_grad():
        logits = m(b["image"].to(dev))
        loss = SegmentationLosses(cuda=True).build_loss("ce")(logits, b["label"].to(dev))
    ref = torch.from_numpy(g["logits"])
    assert logits.shape == ref.shape
    assert rel(logits, ref) < 2e-4
    assert np.array_equal(logits.argmax(1).cpu().numpy(), g["argmax"])
    assert abs(loss.item() - float(g["loss"])) < 1e-3 * float(g["loss"])
    with pytest.raises(ValueError):
        m.train()(b["image"].to(dev))


def test_train_forward_backward_vs_oracle(dev):
    import zs3_oracle as zo
    from zs3_amd.utils.loss import SegmentationLosses
    m, ref = build_pair(tame=True)
    ref64 = copy.deepcopy(ref).double()
    b = zo.make_synthetic_batch(4, 97, seed=7, with_label_emb=False)
    x, y = b["image"], b["label"]
    w = torch.ones(21)
    w[[10, 14]] = 100.0
    m = m.to(dev).train()
    ref.train()
    ref64.train()
    out = m(x.to(dev))
    loss = SegmentationLosses(weight=w.to(dev), cuda=True).build_loss("ce")(out, y.to(dev))
    loss.backward()
    r32 = ref(x)
    l32 = zo.SegmentationLosses(weight=w).build_loss("ce")(r32, y)
    l32.backward()
    r64 = ref64(x.double())
    l64 = zo.SegmentationLosses(weight=w.double()).build_loss("ce")(r64, y)
    l64.backward()
    assert rel(out, r64) < 1e-3
    assert abs(loss.item() - l64.item()) < 1e-4 * abs(l64.item())
    # running statistics (BN batch stats, unbiased running_var, momentum 0.1)
    sd, sd64 = m.state_dict(), ref64.state_dict()
    for k in sd:
        if "running" in k:
            assert rel(sd[k], sd64[k]) < 2e-3, k
        if "num_batches_tracked" in k:
            assert int(sd[k]) == 1
    # gradients: relative L2 error against fp64, judged against the fp32 reference's own error
    bad, rows = [], []
    for (k, p), (_, p32), (_, p64) in zip(m.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
        assert p.grad is not None, k
        g64 = p64.grad
        e = ((p.grad.double().cpu() - g64).norm() / g64.norm().clamp_min(1e-30)).item()
        e32 = ((p32.grad.double() - g64).norm() / g64.norm().clamp_min(1e-30)).item()
        rows.append((e, e32, k))
        if e > GRAD_TOL:
            bad.append((k, e, e32))
    # delivered (printed every run): the worst parameters by absolute error, and by error in units of the fp32 oracle's own
    rows.sort(reverse=True)
    print("\n[train 97x97 B=4] gradient error vs fp64, relative L2 -- worst five: " +
          "; ".join(f"{k} {e:.2e} (oracle fp32 {e32:.1e})" for e, e32, k in rows[:5]))
    worst_ratio = max(rows, key=lambda r: r[0] / max(r[1], 1e-12))
    print(f"[train 97x97 B=4] median {rows[len(rows) // 2][0]:.2e}; largest multiple of the fp32 oracle's own error: "
          f"{worst_ratio[0] / max(worst_ratio[1], 1e-12):.0f}x ({worst_ratio[2]}: {worst_ratio[0]:.2e} vs {worst_ratio[1]:.1e})")
    assert not bad, bad[:10]
    assert m.backbone.conv1.weight.grad.is_contiguous(memory_format=torch.channels_last)


def test_split_forwards_and_state_dict_roundtrip(dev):
    import zs3_oracle as zo
    m, ref = build_pair(num_classes=60, tame=True, sync_bn=True, global_avg_pool_bn=False)
    assert len(m.state_dict()) == 675
    b = zo.make_synthetic_batch(2, 65, num_classes=60, seed=9, with_label_emb=False)
    x = b["image"]
    m = m.to(dev).eval()
    ref.eval()
    with torch.no_grad():
        f4 = m.forward_before_last_conv_finetune(x.to(dev))
        f8 = m.forward_class_last_conv_finetune(f4)
        lg = m.forward_class_prediction(f8, (65, 65))
        top, low = m.backbone(x.to(dev))
        a = m.aspp(top)
        r4 = ref.forward_before_last_conv_finetune(x)
        r8 = ref.forward_class_last_conv_finetune(r4)
        rl = ref.forward_class_prediction(r8, (65, 65))
        rtop, rlow = ref.backbone(x)
        ra = ref.aspp(rtop)
        d1 = m.decoder.forward_class_prediction(f8)
    assert f4.shape == r4.shape and lg.shape == rl.shape and top.shape == rtop.shape and low.shape == rlow.shape
    for got, want in ((f4, r4), (f8, r8), (lg, rl), (top, rtop), (low, rlow), (a, ra), (d1, ref.decoder.forward_class_prediction(r8))):
        assert rel(got, want) < 5e-4
    # state dict written by the product loads into the oracle/reference layout and back
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    ref.load_state_dict(sd)
    m.load_state_dict(ref.state_dict())
    assert m.decoder.last_conv[0].weight.is_contiguous(memory_format=torch.channels_last)
    # train-mode batch of one image fails at the pooled-branch BN exactly like the reference (aspp.py:87)
    m2, _ = build_pair(tame=True)
    m2 = m2.to(dev).train()
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        m2(x[:1].to(dev))


def test_gmmn_mlp_vs_golden(dev, golden):
    from zs3_amd.modeling.gmmn import GMMNnetwork
    g = golden("gmmn_mlp.npz")
    torch.manual_seed(1)
    net = GMMNnetwork(300, 300, 256, 256).to(dev).eval()
    gg = torch.Generator().manual_seed(21)
    emb = torch.randn(37, 300, generator=gg).to(dev).requires_grad_(True)
    z = torch.rand(37, 300, generator=gg).to(dev)
    y = net(emb, z)
    up = torch.randn(37, 256, generator=gg)
    (y * up.to(dev)).sum().backward()
    assert np.allclose(y.detach().cpu().numpy(), g["out"], rtol=1e-3, atol=2e-5)
    names = [str(k) for k in g["grad_names"]]
    for k, r in zip(names, g["grad_stats"]):
        gr = dict(net.named_parameters())[k].grad.double().cpu().reshape(-1)
        assert abs(gr.abs().sum().item() - r[1]) <= 1e-3 * r[1], k
    assert np.allclose(net.model[3].weight.grad.cpu().numpy()[:16, :16], g["grad_w2"], rtol=0, atol=1e-4 * float(np.abs(g["grad_w2"]).max()))
    assert abs(emb.grad.double().abs().sum().item() - g["grad_emb_stats"][1]) <= 1e-3 * g["grad_emb_stats"][1]


def test_supervised_step_vs_oracle(dev):
    """one step of base_trainer.py:16-20 with the fused SGD: updated weights agree with the fp64 oracle"""
    import zs3_oracle as zo
    from zs3_amd.optim import SGD
    from zs3_amd.utils.loss import SegmentationLosses
    m, ref = build_pair(tame=True)
    ref = ref.double()
    init = {k: v.clone() for k, v in ref.state_dict().items()}
    b = zo.make_synthetic_batch(4, 97, seed=11, with_label_emb=False)
    m = m.to(dev).train()
    ref.train()

    def groups(mod, lr):
        return [{"params": mod.get_1x_lr_params(), "lr": lr}, {"params": mod.get_10x_lr_params(), "lr": lr * 10}]

    opt = SGD(groups(m, 1e-3), momentum=0.9, weight_decay=5e-4, nesterov=False)
    opt_r = torch.optim.SGD(groups(ref, 1e-3), momentum=0.9, weight_decay=5e-4, nesterov=False)
    crit, crit_r = SegmentationLosses(cuda=True).build_loss("ce"), zo.SegmentationLosses().build_loss("ce")
    for it in range(2):
        opt.zero_grad()
        loss = crit(m(b["image"].to(dev)), b["label"].to(dev))
        loss.backward()
        opt.step()
        lr_, _ = zo.supervised_step(ref, opt_r, crit_r, b["image"].double(), b["label"])
        assert abs(loss.item() - lr_) < 2e-4 * abs(lr_), (it, loss.item(), lr_)
    sd, sdr = m.state_dict(), ref.state_dict()
    for k in ("decoder.pred_conv.weight", "decoder.pred_conv.bias", "decoder.last_conv.4.weight", "aspp.conv1.weight",
              "backbone.layer4.2.conv3.weight", "backbone.layer1.0.conv1.weight", "backbone.conv1.weight", "backbone.bn1.weight"):
        d, dr = sd[k].double().cpu() - init[k], sdr[k] - init[k]
        e = ((d - dr).norm() / dr.norm()).item()
        print(f"[supervised step] update of {k}: rel L2 {e:.2e}")
        assert e < UPDATE_TOL[k], k


def test_context_60_classes_forward_and_step(dev):
    """BASELINE configs[3]/[4] shape of the head: 60 classes (datasets/context.py:22) with global_avg_pool_bn=False and
    labels up to 59 + ignore 255: eval logits / argmax and one supervised training step against the oracle"""
    import zs3_oracle as zo
    from zs3_amd.optim import SGD
    from zs3_amd.utils.loss import SegmentationLosses
    m, ref = build_pair(num_classes=60, tame=True, global_avg_pool_bn=False)
    assert len(m.state_dict()) == 675
    g = torch.Generator().manual_seed(60)
    image = torch.randn(2, 3, 65, 65, generator=g)
    label = torch.randint(0, 60, (2, 9, 9), generator=g).float().repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :65, :65]
    label = torch.nn.functional.pad(label, (0, 65 - label.shape[2], 0, 65 - label.shape[1]), value=255.0)
    label[:, :3] = 255
    m = m.to(dev).eval()
    ref.eval()
    with torch.no_grad():
        out, out_r = m(image.to(dev)), ref(image)
    assert out.shape == (2, 60, 65, 65) and rel(out, out_r) < 2e-4
    top2 = out_r.topk(2, dim=1).values
    sure = (top2[:, 0] - top2[:, 1]) > 2e-4 * out_r.abs().max()
    assert torch.equal(out.argmax(1).cpu()[sure], out_r.argmax(1)[sure]) and sure.float().mean() > 0.99
    m.train()
    ref = ref.double().train()

    def groups(mod, lr):
        return [{"params": mod.get_1x_lr_params(), "lr": lr}, {"params": mod.get_10x_lr_params(), "lr": lr * 10}]

    opt = SGD(groups(m, 1e-3), momentum=0.9, weight_decay=5e-4, nesterov=False)
    opt_r = torch.optim.SGD(groups(ref, 1e-3), momentum=0.9, weight_decay=5e-4, nesterov=False)
    w = torch.ones(60)
    w[[5, 17]] = 100.0
    init = ref.decoder.pred_conv.weight.detach().clone()
    opt.zero_grad()
    loss = SegmentationLosses(weight=w.to(dev), cuda=True).build_loss("ce")(m(image.to(dev)), label.to(dev))
    loss.backward()
    opt.step()
    loss_r, _ = zo.supervised_step(ref, opt_r, zo.SegmentationLosses(weight=w.double()).build_loss("ce"), image.double(), label)
    assert abs(loss.item() - loss_r) < 2e-4 * abs(loss_r), (loss.item(), loss_r)
    d = m.decoder.pred_conv.weight.detach().double().cpu() - init
    dr = ref.decoder.pred_conv.weight.detach() - init
    assert ((d - dr).norm() / dr.norm()).item() < 3e-2


def test_gmmn_step_vs_oracle(dev):
    """train_pascal_GMMN.py:139-268 with the reference's CPU noise stream: per-step G/C losses and the updated
    generator / pred_conv agree with the oracle; the backbone is untouched; BN running stats moved."""
    import zs3_oracle as zo
    from zs3_amd.gmmn_trainer import GMMNStep
    from zs3_amd.modeling.gmmn import GMMNnetwork
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    seen = [c for c in range(21) if c not in (10, 14)]
    m, ref = build_pair(tame=True)
    torch.manual_seed(2)
    gen = GMMNnetwork(300, 300, 256, 256)
    gen_r = zo.GMMNnetwork(300, 300, 256, 256)
    gen_r.load_state_dict(gen.state_dict())
    gen.model[2].p = 0.0
    gen_r.model[2].p = 0.0
    m, gen = m.to(dev).train(), gen.to(dev).train()
    ref.train()
    gen_r.train()
    w = torch.ones(21)
    w[[10, 14]] = 100.0

    def groups(mod, lr):
        return [{"params": mod.get_1x_lr_params(), "lr": lr}, {"params": mod.get_10x_lr_params(), "lr": lr * 10}]

    opt, opt_g = SGD(groups(m, 0.007), momentum=0.9, weight_decay=5e-4), Adam(gen.parameters(), lr=2e-4)
    opt_r, opt_gr = torch.optim.SGD(groups(ref, 0.007), momentum=0.9, weight_decay=5e-4), torch.optim.Adam(gen_r.parameters(), lr=2e-4)
    step = GMMNStep(m, gen, opt, opt_g, SegmentationLosses(weight=w.to(dev), cuda=True).build_loss("ce"), seen=seen,
                    unseen=[10, 14], noise="cpu")
    stem0 = m.backbone.conv1.weight.detach().clone()
    rm0 = m.backbone.bn1.running_mean.clone()
    for it in range(2):
        b = zo.make_synthetic_batch(4, 65, seed=200 + it, with_label_emb=True)
        torch.manual_seed(13 + it)
        gl_r, cl_r = zo.gmmn_step(ref, gen_r, opt_r, opt_gr, zo.SegmentationLosses(weight=w).build_loss("ce"),
                                  zo.GMMNLoss().build_loss(), b["image"], b["label"], b["label_emb"], seen=seen, unseen=[10, 14])
        torch.manual_seed(13 + it)
        gl, cl, out = step(b["image"].to(dev), b["label"].to(dev), b["label_emb"].to(dev))
        assert abs(gl - gl_r) < 2e-3 * abs(gl_r), (it, gl, gl_r)
        assert abs(cl - cl_r) < 1e-3 * abs(cl_r), (it, cl, cl_r)
        assert out.shape == (4, 21, 65, 65)
    for (k, p), (_, pr) in zip(gen.named_parameters(), gen_r.named_parameters()):
        # Adam moves every weight by ~lr per step whatever the gradient magnitude: where a gradient is ~0 its sign,
        # hence the step, is rounding noise.  ~50 Adam steps were taken: bound the worst element by a few steps and
        # the mean error tightly.
        e_max = rel(p, pr)
        e_mean = ((p.detach().cpu() - pr.detach()).abs().mean() / pr.detach().abs().mean()).item()
        print(f"[gmmn step] generator {k}: max {e_max:.2e} mean {e_mean:.2e}")
        assert e_max < 2e-2, k          # delivered 1.3e-2 (model.0.bias): one Adam step (2e-4) of a sign flip on |w| ~ 1e-2
        assert e_mean < 1.5e-3, k       # 3 x delivered (5.1e-4)
    assert rel(m.decoder.pred_conv.weight, ref.decoder.pred_conv.weight) < 2e-3
    assert rel(m.decoder.pred_conv.bias, ref.decoder.pred_conv.bias) < 2e-3
    assert torch.equal(m.backbone.conv1.weight.detach(), stem0)           # backbone receives no gradient
    assert not torch.equal(m.backbone.bn1.running_mean, rm0)              # but BN statistics drift (train() mode)
    assert rel(m.backbone.bn1.running_mean, ref.backbone.bn1.running_mean) < 1e-3


@pytest.mark.parametrize("context_aware,avg_feat", [(False, False), (True, True)])
def test_gcn_context_step_vs_oracle(dev, context_aware, avg_feat):
    """train_context_GMMN_GCNcontext.py:239-457 with the reference's CPU noise stream: per-step generator,
    graph-generator and classifier losses and the updated generators / pred_conv agree with the oracle (which reproduces
    the reference's own trajectory bit for bit, tests/test_oracle_golden.py)."""
    import zs3_oracle as zo
    from zs3_amd.gcn_trainer import GCNContextStep
    from zs3_amd.modeling.gmmn import GMMNnetwork, GMMNnetwork_GCN
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    seen = [c for c in range(21) if c not in (10, 14)]
    m, ref = build_pair(tame=True)
    torch.manual_seed(2)
    gen, gcn = GMMNnetwork(300, 300, 256, 256), GMMNnetwork_GCN(300, 300, 256, 256)
    gen_r, gcn_r = zo.GMMNnetwork(300, 300, 256, 256), zo.GMMNnetwork_GCN(300, 300, 256, 256)
    gen_r.load_state_dict(gen.state_dict())
    gcn_r.load_state_dict(gcn.state_dict())
    for net in (gen, gen_r):
        net.model[2].p = 0.0
    for net in (gcn, gcn_r):
        net.dropout.p = 0.0
    m, gen, gcn = m.to(dev).train(), gen.to(dev).train(), gcn.to(dev).train()
    ref.train(), gen_r.train(), gcn_r.train()
    w = torch.ones(21)
    w[[10, 14]] = 100.0

    def groups(mod, lr):
        return [{"params": mod.get_1x_lr_params(), "lr": lr}, {"params": mod.get_10x_lr_params(), "lr": lr * 10}]

    opt, opt_g, opt_c = SGD(groups(m, 0.007), momentum=0.9, weight_decay=5e-4), Adam(gen.parameters(), lr=2e-4), \
        Adam(gcn.parameters(), lr=2e-4)
    opt_r = torch.optim.SGD(groups(ref, 0.007), momentum=0.9, weight_decay=5e-4)
    opt_gr, opt_cr = torch.optim.Adam(gen_r.parameters(), lr=2e-4), torch.optim.Adam(gcn_r.parameters(), lr=2e-4)
    step = GCNContextStep(m, gen, gcn, opt, opt_g, opt_c, SegmentationLosses(weight=w.to(dev), cuda=True).build_loss("ce"),
                          seen=seen, unseen=[10, 14], noise="cpu", GCN_weight=0.1, GCN_avg_feat=avg_feat,
                          context_aware=context_aware)
    for it in range(2):
        b = zo.make_synthetic_batch(4, 65, seed=500 + it, with_label_emb=True)
        torch.manual_seed(31 + it)
        gl_r, gcl_r, cl_r = zo.gcn_context_step(ref, gen_r, gcn_r, opt_r, opt_gr, opt_cr,
                                                zo.SegmentationLosses(weight=w).build_loss("ce"), zo.GMMNLoss().build_loss(),
                                                b["image"], b["label"], b["label_emb"], seen=seen, unseen=[10, 14],
                                                gcn_weight=0.1, gcn_avg_feat=avg_feat, context_aware=context_aware)
        torch.manual_seed(31 + it)
        gl, gcl, cl, out = step(b["image"].to(dev), b["label"].to(dev), b["label_emb"].to(dev))
        assert step.last_num_clusters > 100           # 4 images x (up to 49 cells + border) clusters
        assert abs(gl - gl_r) < 2e-3 * abs(gl_r), (it, gl, gl_r)
        assert abs(gcl - gcl_r) < 2e-3 * abs(gcl_r), (it, gcl, gcl_r)
        assert abs(cl - cl_r) < 1e-3 * abs(cl_r), (it, cl, cl_r)
        assert out.shape == (4, 21, 65, 65)
    for net, net_r in ((gen, gen_r), (gcn, gcn_r)):
        for (k, p), (_, pr) in zip(net.named_parameters(), net_r.named_parameters()):
            # Adam: see test_gmmn_step_vs_oracle -- an element whose gradient is ~0 moves by +-lr = 2e-4 per update on the
            # sign of rounding noise, whatever its magnitude.  The graph generator took 6 updates (3 trainable images x 2
            # steps); its 0.01-sized biases (bias gradient = column sum of the MMD gradient, which nearly cancels) get the
            # absolute bound of those 6 updates, everything else the relative bounds of the GMMN test.
            diff = (p.detach().cpu() - pr.detach()).abs()
            if net is gcn and k.endswith("bias"):
                assert diff.max().item() <= 6 * 2e-4 * 1.01 and diff.mean().item() < 3 * 2e-4, (k, diff.max().item())
            else:
                assert diff.max().item() < 2e-2 * pr.detach().abs().max().item(), (k, diff.max().item())
                assert (diff.mean() / pr.detach().abs().mean()).item() < 2e-3, k
    assert rel(m.decoder.pred_conv.weight, ref.decoder.pred_conv.weight) < 2e-3
    assert rel(m.decoder.pred_conv.bias, ref.decoder.pred_conv.bias) < 2e-3


def test_gradient_accumulation_over_two_backwards(dev):
    """a second backward before the optimizer step adds into the existing .grad tensors.  Weight gradients are produced on
    a side stream, and in this case they are read (accumulated) inside the backward pass instead of after its
    end-of-backward join: the sum must equal the separately computed gradients exactly."""
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.utils.loss import SegmentationLosses
    from zs3_amd.utils.synthetic import make_batch
    torch.manual_seed(3)
    m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, nn.BatchNorm2d):
            mod.momentum = 0.0          # keep the running statistics fixed: the three passes see the same model
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    b1, b2 = make_batch(4, 193, seed=11, device=dev), make_batch(4, 193, seed=12, device=dev)
    single = []
    for b in (b1, b2):
        for p in m.parameters():
            p.grad = None
        crit(m(b["image"]), b["label"]).backward()
        single.append([p.grad.clone() for p in m.parameters()])
    for p in m.parameters():
        p.grad = None
    crit(m(b1["image"]), b1["label"]).backward()
    crit(m(b2["image"]), b2["label"]).backward()
    torch.cuda.synchronize()
    for (name, p), g1, g2 in zip(m.named_parameters(), *single):
        assert torch.equal(p.grad, g1 + g2), name


def _default_init_train_step(dev, golden):
    """One train-mode step of the default-initialised network (seed 1, no taming, dropout off, 65x65, B=2) on the GPU, measured
    (a) against the reference's own fp32 outputs (tests/golden/deeplab_forward.npz, written by tools/make_goldens.py from
    /root/reference) and (b) against the fp64 oracle, next to the reference's own fp32-vs-fp64 error."""
    import zs3_oracle as zo
    from zs3_amd.utils.loss import SegmentationLosses
    g = golden("deeplab_forward.npz")
    m, ref = build_pair(tame=False)
    ref64 = copy.deepcopy(ref).double().train()
    b = zo.make_synthetic_batch(2, 65, seed=7, with_label_emb=False)
    w = torch.ones(21)
    w[[10, 14]] = 100.0
    m = m.to(dev).train()
    out = m(b["image"].to(dev))
    loss = SegmentationLosses(weight=w.to(dev), cuda=True).build_loss("ce")(out, b["label"].to(dev))
    loss.backward()
    torch.cuda.synchronize()
    r64 = ref64(b["image"].double())
    zo.SegmentationLosses(weight=w.double()).build_loss("ce")(r64, b["label"]).backward()
    gold = torch.from_numpy(g["train_logits"])
    gp, gs = torch.from_numpy(g["grad_pred_w"]), torch.from_numpy(g["grad_stem_w"])
    e = {"logits": rel(out, gold), "loss": abs(loss.item() - float(g["train_loss"])) / abs(float(g["train_loss"])),
         "pred": rel(m.decoder.pred_conv.weight.grad, gp), "stem": rel(m.backbone.conv1.weight.grad[:8], gs)}
    sd = m.state_dict()
    e["run"], e["run_key"] = 0.0, None
    for k, r in zip(g["run_names"], g["run_stats"]):
        t = sd[str(k)].double().cpu().reshape(-1)
        d = abs(t.abs().sum().item() - r[1]) / max(abs(r[1]), 1e-30)
        if d > e["run"]:
            e["run"], e["run_key"] = d, str(k)
    # against fp64: ours, and the reference's own fp32 result
    e["ref_logits"], e["ref_pred"] = rel(gold, r64), rel(gp, ref64.decoder.pred_conv.weight.grad)
    e["ref_stem"] = rel(gs, ref64.backbone.conv1.weight.grad[:8])
    e["our_logits"], e["our_pred"] = rel(out, r64), rel(m.decoder.pred_conv.weight.grad, ref64.decoder.pred_conv.weight.grad)
    e["our_stem"] = rel(m.backbone.conv1.weight.grad[:8], ref64.backbone.conv1.weight.grad[:8])
    return e


def _report(tag, e):
    print(f"[default-init train, {tag}] vs reference goldens: logits {e['logits']:.2e} loss {e['loss']:.2e} grad_pred_w "
          f"{e['pred']:.2e} grad_stem_w {e['stem']:.2e} running stats {e['run']:.2e} ({e['run_key']}); vs fp64: logits "
          f"{e['our_logits']:.2e} (reference fp32: {e['ref_logits']:.2e} -> {e['our_logits'] / e['ref_logits']:.1f}x), grad_pred_w "
          f"{e['our_pred']:.2e} ({e['ref_pred']:.2e} -> {e['our_pred'] / e['ref_pred']:.1f}x), grad_stem_w {e['our_stem']:.2e} "
          f"({e['ref_stem']:.2e} -> {e['our_stem'] / e['ref_stem']:.1f}x)")


def test_default_init_train_step_vs_reference_goldens(dev, golden):
    """The reference's OWN train-mode outputs for the default-initialised network: logits, loss, the classifier and stem weight
    gradients and the 226 BatchNorm running statistics -- the goldens no other GPU test touches -- on the product's default
    arithmetic: forward convolutions f16x3 (fp16 hi/lo operands, 2^-22-class products), data / weight gradients bf16x3.

    This configuration is ill-conditioned: batch statistics over 2 x 5 x 5 positions in 101 layers amplify a rounding
    difference ~1e4-fold -- the reference's own fp32 result sits 1.7e-3 (logits) to 6e-2 (stem gradient) away from its fp64
    evaluation (measured below, not assumed), so two correct fp32-class evaluations differ by about that much.  The amplification
    is all in the FORWARD pass (the backward is linear in the forward's activations): with bf16x3 forward products (2^-16-class,
    rounds 1-3) the same step sat 3.6e-2 from the goldens = 22x / 31x / 5.6x the reference's own error (next test); with the
    fp16 split it sits where a second fp32 evaluation would (tools/probe/split_emulation.py predicted 1.3x / 1.6x / 1.4x on the CPU).
    Delivered: logits 2.0e-3 from the goldens (bf16 split: 3.6e-2; the exact-fp32 instantiation: 2.1e-3), loss 3.0e-5, classifier
    gradient 1.6e-3, running statistics 8.4e-6; against fp64 0.8x / 1.3x / 0.9x the reference's own fp32 error (logits / classifier
    gradient / stem gradient) -- where the exact-fp32 instantiation sits (0.8x / 0.9x / 1.0x).  (Without the 2^6 weight scale of
    the fp16 plane -- common.h: a typical weight's lo half is an fp16 subnormal otherwise -- 3.2e-3 and 1.8x / 2.1x / 1.4x.)
    Assertions: against the fp64 oracle at most 2.5x the reference's own fp32 error for logits and both gradients; against the
    goldens 2.5x that error as well (each evaluation is ~1 such error from fp64); loss and running statistics at 3x delivered."""
    e = _default_init_train_step(dev, golden)
    _report("f16x3 forward, bf16x3 backward (default)", e)
    assert e["our_logits"] < 2.5 * e["ref_logits"]
    assert e["our_pred"] < 2.5 * e["ref_pred"]
    assert e["our_stem"] < 2.5 * e["ref_stem"]
    assert e["logits"] < 2.5 * e["ref_logits"]
    assert e["pred"] < 2.5 * e["ref_pred"]
    assert e["loss"] < 1e-4          # 3 x delivered (3.0e-5)
    assert e["run"] < 3e-5           # 3 x delivered (8.4e-6)


def test_default_init_train_step_with_bf16x3_forward(dev, golden, monkeypatch):
    """The same step with the forward products on the bf16 split (ZS3_FWD_F16=0, the arithmetic of rounds 1-3): kept as the
    measured comparison.  Delivered: logits 3.6e-2 from the goldens, loss 3.6e-4, classifier gradient 3.5e-2, running
    statistics 2.5e-3; against fp64 22x / 31x / 5.6x the reference's own fp32 error.  Bounds: 3x delivered, and 64x / 64x / 16x."""
    from zs3_amd import ops
    monkeypatch.setattr(ops, "FWD_F16", False)
    e = _default_init_train_step(dev, golden)
    _report("bf16x3 forward and backward", e)
    assert e["logits"] < 0.11        # 3 x delivered (3.6e-2)
    assert e["loss"] < 1.1e-3        # 3 x delivered (3.6e-4); the north-star 1e-3 is met
    assert e["pred"] < 0.11          # 3 x delivered (3.5e-2)
    assert e["run"] < 7.5e-3         # 3 x delivered (2.5e-3)
    assert e["our_logits"] < 64 * e["ref_logits"]
    assert e["our_pred"] < 64 * e["ref_pred"]
    assert e["our_stem"] < 16 * e["ref_stem"]
    assert e["our_logits"] > 4 * e["ref_logits"]     # (and it IS the forward products: this run does not meet the default's bound)


def test_default_init_train_step_in_exact_fp32_matches_the_references_own_error(dev, golden):
    """VERDICT r3 #3a: the same ill-conditioned step with every convolution product on v_mfma_f32_32x32x2_f32
    (ops.set_exact_fp32: prec = 0 of the register-staged conv / wgrad kernels, fp32 weight planes; everything else -- BatchNorm
    kernels, epilogues, pooling, resize, loss, autograd wiring -- is the product path unchanged).  If the 3.6e-2 of the bf16x3
    run were a structural error it would survive the change of arithmetic; it does not: the result sits where the reference's
    own fp32 run sits relative to fp64 (the two differ in summation order only).  Bounds: 4x the reference's own fp32 error
    against fp64 for logits and both gradients; against the goldens 4x that error as well (two fp32 evaluations of a chaotic map
    are each ~1 such error from fp64)."""
    from zs3_amd import ops
    ops.set_exact_fp32(True)
    try:
        e = _default_init_train_step(dev, golden)
    finally:
        ops.set_exact_fp32(False)
    _report("exact fp32", e)
    assert e["our_logits"] < 4 * e["ref_logits"]
    assert e["our_pred"] < 4 * e["ref_pred"]
    assert e["our_stem"] < 4 * e["ref_stem"]
    assert e["logits"] < 4 * e["ref_logits"]
    assert e["pred"] < 4 * e["ref_pred"]
    assert e["loss"] < 1e-4
