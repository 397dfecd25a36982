"""Pins the CPU oracle (oracle/zs3_oracle) against golden vectors produced by the reference itself
(tools/make_goldens.py imports /root/reference).  CPU only; no reference access at test time."""
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

import zs3_oracle as zo


def stats(t):
    t = t.detach().double().reshape(-1)
    return np.array([t.sum().item(), t.abs().sum().item(), t[0].item(), t[-1].item()])


def check_table(named, names, ref, rtol, atol=1e-6, what=""):
    got = dict(named)
    assert list(names) == [k for k in got] or set(names) == set(got), f"{what}: key mismatch"
    for k, r in zip(names, ref):
        s = stats(got[str(k)])
        scale = max(abs(r[1]), 1e-30)  # the plain sum cancels: judge it against the abs-sum
        ok = abs(s[1] - r[1]) <= rtol * scale + atol and abs(s[0] - r[0]) <= rtol * scale + atol
        ok = ok and np.allclose(s[2:], r[2:], rtol=max(rtol, 1e-12) * 10, atol=atol + rtol)
        assert ok, f"{what}:{k}: {s} vs {r}"


@pytest.fixture(scope="module")
def torch_threads():
    torch.set_num_threads(min(8, torch.get_num_threads()))


def test_constructor_init_matches_reference(golden, torch_threads):
    g = golden("init.npz")
    torch.manual_seed(1)
    m = zo.DeepLab(num_classes=21, pretrained=False, sync_bn=False)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["names21"]]
    assert len(sd) == 680
    check_table(sd.items(), g["names21"], g["stats21"], rtol=0, atol=0, what="init21")
    assert sum(p.numel() for p in m.get_1x_lr_params()) == int(g["lr1"]) == 42500160
    assert sum(p.numel() for p in m.get_10x_lr_params()) == int(g["lr10"]) == 16844149
    torch.manual_seed(1)
    m60 = zo.DeepLab(num_classes=60, pretrained=False, sync_bn=True, global_avg_pool_bn=False)
    sd = m60.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["names60"]] and len(sd) == 675
    check_table(sd.items(), g["names60"], g["stats60"], rtol=0, atol=0, what="init60")
    torch.manual_seed(1)
    gen = zo.GMMNnetwork(300, 300, 256, 256)
    assert list(gen.state_dict().keys()) == [str(k) for k in g["names_g"]]
    check_table(gen.state_dict().items(), g["names_g"], g["stats_g"], rtol=0, atol=0, what="gmmn")
    torch.manual_seed(3)
    gen2 = zo.GMMNnetwork(300, 300, 0, 256, semantic_reconstruction=True)
    check_table(gen2.state_dict().items(), g["names_g2"], g["stats_g2"], rtol=0, atol=0, what="gmmn2")


def test_oracle_at_513_at_the_default_crop_and_at_output_stride_8(golden, torch_threads):
    """tests/golden/sizes.npz (tools/make_goldens.py `sizes`): the reference at 513 x 513, at its default crop 312 x 312
    (train_pascal.py:203-204; eval and one train-mode forward + weighted CE + backward) and with output_stride = 8
    (resnet.py:72-74, aspp.py:49-50).  tests/test_gpu_parity_sizes.py judges the HIP path with the oracle at exactly these
    configurations: this is what pins it there."""
    g = golden("sizes.npz")
    for size in (513, 312):
        torch.manual_seed(1)
        m = zo.DeepLab(num_classes=21, pretrained=False).eval()
        x = zo.make_synthetic_batch(2, size, seed=size, with_label_emb=False)["image"]
        assert np.allclose(stats(x), g[f"eval{size}_in_stats"], rtol=0, atol=0)
        with torch.no_grad():
            logits = m(x)
        ref = torch.from_numpy(g[f"eval{size}_logits_sub"])
        assert (logits[:, :, ::8, ::8] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
        assert np.allclose(stats(logits), g[f"eval{size}_logits_stats"], rtol=1e-5)
        assert np.array_equal(logits.argmax(1).numpy().astype(np.uint8), g[f"eval{size}_argmax"])
    torch.manual_seed(1)
    m = zo.DeepLab(num_classes=21, pretrained=False)
    for name, mod in m.named_modules():
        if name.endswith("bn3"):
            mod.weight.data.fill_(0.1)
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    m.train()
    b = zo.make_synthetic_batch(2, 312, seed=1312, with_label_emb=False)
    w = torch.ones(21)
    w[[10, 14]] = 100.0
    logits = m(b["image"])
    loss = zo.SegmentationLosses(weight=w).build_loss("ce")(logits, b["label"])
    loss.backward()
    ref = torch.from_numpy(g["train312_logits_sub"])
    assert (logits.detach()[:, :, ::8, ::8] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert abs(loss.item() - float(g["train312_loss"])) <= 1e-6 * abs(float(g["train312_loss"]))
    check_table(((k, p.grad) for k, p in m.named_parameters()), g["train312_grad_names"], g["train312_grad_stats"], rtol=2e-3,
                what="grad312")
    check_table(((k, v) for k, v in m.state_dict().items() if "running" in k), g["train312_run_names"], g["train312_run_stats"],
                rtol=1e-5, what="running312")
    gp = torch.from_numpy(g["train312_grad_pred_w"])
    assert (m.decoder.pred_conv.weight.grad - gp).abs().max().item() <= 1e-4 * gp.abs().max().item()
    torch.manual_seed(1)
    m = zo.DeepLab(output_stride=8, num_classes=21, pretrained=False)
    for name, mod in m.named_modules():
        if name.endswith("bn3"):
            mod.weight.data.fill_(0.1)
    m.eval()
    x = zo.make_synthetic_batch(2, 65, seed=8, with_label_emb=False)["image"]
    with torch.no_grad():
        logits = m(x)
        top, low = m.backbone(x)
    ref = torch.from_numpy(g["os8_logits"])
    assert tuple(top.shape) == tuple(int(v) for v in g["os8_top_shape"]) == (2, 2048, 9, 9)
    assert (logits - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert np.array_equal(logits.argmax(1).numpy().astype(np.uint8), g["os8_argmax"])
    assert np.allclose(stats(top), g["os8_top_stats"], rtol=1e-5) and np.allclose(stats(low), g["os8_low_stats"], rtol=1e-5)


def test_deeplab_forward_backward_matches_reference(golden, torch_threads):
    g = golden("deeplab_forward.npz")
    torch.manual_seed(1)
    m = zo.DeepLab(num_classes=21, pretrained=False)
    b = zo.make_synthetic_batch(2, 65, seed=7, with_label_emb=False)
    x, y = b["image"], b["label"]
    assert np.allclose(stats(x), g["in_stats"], rtol=0, atol=0) and np.allclose(stats(y), g["label_stats"])
    m.eval()
    with torch.no_grad():
        logits = m(x)
        feat = m.forward_before_class_prediction(x)
        top, low = m.backbone(x)
    ref = torch.from_numpy(g["eval_logits"])
    assert (logits - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert np.array_equal(logits.argmax(1).numpy().astype(np.uint8), g["eval_argmax"])
    assert np.allclose(feat[:, :8, ::4, ::4].numpy(), g["eval_feat_slice"], rtol=1e-4, atol=1e-5)
    assert np.allclose(stats(top), g["eval_top_stats"], rtol=1e-5) and np.allclose(stats(low), g["eval_low_stats"], rtol=1e-5)
    # train mode with dropout disabled
    m.train()
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    w = torch.ones(21)
    w[[10, 14]] = 100.0
    logits = m(x)
    loss = zo.SegmentationLosses(weight=w).build_loss("ce")(logits, y)
    loss.backward()
    ref = torch.from_numpy(g["train_logits"])
    assert (logits.detach() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    assert abs(loss.item() - float(g["train_loss"])) <= 1e-5 * abs(float(g["train_loss"]))
    gp = torch.from_numpy(g["grad_pred_w"])
    assert (m.decoder.pred_conv.weight.grad - gp).abs().max() <= 1e-3 * gp.abs().max()
    gs = torch.from_numpy(g["grad_stem_w"])
    assert (m.backbone.conv1.weight.grad[:8] - gs).abs().max() <= 5e-3 * gs.abs().max()
    runs = {k: v for k, v in m.state_dict().items() if "running" in k}
    check_table(runs.items(), g["run_names"], g["run_stats"], rtol=1e-4, atol=1e-5, what="running")
    lg = logits.detach()
    for mode, key in (("focal", "loss_focal"), ("ce_finetune", "loss_ce_finetune")):
        v = zo.SegmentationLosses(weight=w).build_loss(mode)(lg, y).item()
        assert abs(v - float(g[key])) <= 2e-5 * abs(float(g[key])), mode
    v = zo.SegmentationLosses().build_loss("ce")(lg, y).item()
    assert abs(v - float(g["loss_ce_noweight"])) <= 2e-5 * abs(float(g["loss_ce_noweight"]))
    for wt in (None, w):
        a = zo.cross_entropy_2d(lg, y, wt).item()
        b = zo.cross_entropy_2d_closed_form(lg, y, wt).item()
        assert abs(a - b) <= 2e-6 * abs(a)
    m.eval()
    with torch.no_grad():
        f4 = m.forward_before_last_conv_finetune(x)
        f8 = m.forward_class_last_conv_finetune(f4)
        lg2 = m.forward_class_prediction(f8, (65, 65))
    for t, key in ((f4, "split_f4_stats"), (f8, "split_f8_stats"), (lg2, "split_logits_stats")):
        assert np.allclose(stats(t), g[key], rtol=2e-4, atol=1e-4), key


def test_baseline_config0_forward_and_loss(golden, torch_threads):
    """BASELINE.json configs[0] (the reference's own CPU-runnable case): forward + CE loss on one random 3x129x129 tensor"""
    g = golden("config0_129.npz")
    torch.manual_seed(1)
    m = zo.DeepLab(num_classes=21, pretrained=False).eval()
    b = zo.make_synthetic_batch(1, 129, seed=129, with_label_emb=False)
    assert np.allclose(stats(b["image"]), g["in_stats"], rtol=1e-12) and np.allclose(stats(b["label"]), g["label_stats"], rtol=1e-12)
    with torch.no_grad():
        logits = m(b["image"])
        loss = zo.SegmentationLosses().build_loss("ce")(logits, b["label"])
    ref = torch.from_numpy(g["logits"])
    assert ((logits - ref).abs().max() / ref.abs().max()).item() < 1e-6
    assert np.array_equal(logits.argmax(1).numpy(), g["argmax"])
    assert abs(loss.item() - float(g["loss"])) < 1e-6 * float(g["loss"])
    assert bool(g["train_b1_raises"])
    with pytest.raises(ValueError):          # aspp.py:87: the pooled-branch BN sees one value per channel
        m.train()(b["image"])


def test_supervised_trajectory_matches_reference(golden, torch_threads):
    """11 SGD iterations of base_trainer.py:5-25 at 65x65, B=2, dropout active (CPU RNG stream identical)."""
    g = golden("supervised_traj.npz")
    torch.manual_seed(1)
    m = zo.DeepLab(num_classes=21, pretrained=False)
    init = {k: v.clone() for k, v in m.state_dict().items()}
    params = [{"params": m.get_1x_lr_params(), "lr": 1e-5}, {"params": m.get_10x_lr_params(), "lr": 1e-4}]
    opt = torch.optim.SGD(params, momentum=0.9, weight_decay=5e-4, nesterov=False)
    crit = zo.SegmentationLosses().build_loss("ce")
    m.train()
    torch.manual_seed(11)
    losses = []
    for it in range(11):
        b = zo.make_synthetic_batch(2, 65, seed=100 + it, with_label_emb=False)
        zo.apply_lr(opt, zo.poly_lr(1e-5, it, 0, 11, 2))
        loss, _ = zo.supervised_step(m, opt, crit, b["image"], b["label"])
        losses.append(loss)
    assert np.allclose(losses, g["losses"], rtol=1e-5), (losses, g["losses"])    # exact (0.0) in this container
    assert np.allclose([pg["lr"] for pg in opt.param_groups], g["final_lr"], rtol=1e-12)
    assert int(m.state_dict()["backbone.bn1.num_batches_tracked"]) == int(g["nbt"]) == 11
    sd = m.state_dict()
    for k, r in zip(g["delta_names"], g["delta_stats"]):
        k = str(k)
        d = stats(sd[k] - init[k])
        assert abs(d[1] - r[1]) <= 0.01 * r[1] + 1e-12, (k, d, r)  # sum |w_final - w_init|


def test_mmd_matches_reference(golden):
    g = golden("mmd.npz")
    crit = zo.GMMNLoss().build_loss()
    for name in ("rand128", "far128", "small4", "near128"):
        gen = torch.from_numpy(g[f"{name}_gen"]).requires_grad_(True)
        real = torch.from_numpy(g[f"{name}_real"])
        loss = crit(gen, real)
        loss.backward()
        ref = float(g[f"{name}_loss"])
        assert abs(loss.item() ** 2 - ref ** 2) <= 1e-6, name  # tolerance is absolute on loss^2 (cancellation)
        rg = torch.from_numpy(g[f"{name}_grad"])
        assert (gen.grad - rg).abs().max() <= 2e-3 * rg.abs().max() + 1e-7, name
    real = torch.randn(128, 256, generator=torch.Generator().manual_seed(5))
    assert crit(real.clone(), real).item() == float(g["identical_loss"]) == 0.0


def test_gmmn_mlp_matches_reference(golden):
    g = golden("gmmn_mlp.npz")
    torch.manual_seed(1)
    net = zo.GMMNnetwork(300, 300, 256, 256).eval()
    gg = torch.Generator().manual_seed(21)
    emb = torch.randn(37, 300, generator=gg).requires_grad_(True)
    z = torch.rand(37, 300, generator=gg)
    y = net(emb, z)
    up = torch.randn(37, 256, generator=gg)
    assert np.array_equal(up.numpy(), g["up"])
    (y * up).sum().backward()
    assert np.allclose(y.detach().numpy(), g["out"], rtol=1e-5, atol=1e-6)
    check_table(((k, p.grad) for k, p in net.named_parameters()), g["grad_names"], g["grad_stats"], rtol=1e-4, atol=1e-4)
    assert np.allclose(stats(emb.grad), g["grad_emb_stats"], rtol=1e-4, atol=1e-4)
    assert np.allclose(net.model[3].weight.grad.numpy()[:16, :16], g["grad_w2"], rtol=1e-4, atol=1e-5)


def test_gmmn_trajectory_matches_reference(golden, torch_threads):
    """11 iterations of train_pascal_GMMN.py:139-268 at 65x65, B=4 (image 3 holds an unseen class).  In this container
    the oracle reproduces the reference's trajectory exactly (difference 0.0 in every logged loss); the tolerances leave
    room for another CPU / thread count."""
    g = golden("gmmn_traj.npz")
    seen = [c for c in range(21) if c not in (10, 14)]
    torch.manual_seed(1)
    m = zo.DeepLab(num_classes=21, pretrained=False)
    gen = zo.GMMNnetwork(300, 300, 256, 256)
    params = [{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}]
    opt = torch.optim.SGD(params, momentum=0.9, weight_decay=5e-4, nesterov=False)
    opt_g = torch.optim.Adam(gen.parameters(), lr=2e-4)
    w = torch.ones(21)
    w[[10, 14]] = 100.0
    crit = zo.SegmentationLosses(weight=w).build_loss("ce")
    mmd = zo.GMMNLoss().build_loss()
    m.train()
    gen.train()
    torch.manual_seed(13)
    closs, gloss = [], []
    for it in range(11):
        b = zo.make_synthetic_batch(4, 65, seed=200 + it, with_label_emb=True)
        zo.apply_lr(opt, zo.poly_lr(0.007, it, 0, 11, 2))
        gl, cl = zo.gmmn_step(m, gen, opt, opt_g, crit, mmd, b["image"], b["label"], b["label_emb"], seen=seen,
                              unseen=[10, 14])
        closs.append(cl)
        gloss.append(gl)
    assert np.allclose(closs, g["closs"], rtol=1e-5), (closs, g["closs"])
    assert np.allclose(gloss, g["gloss"], rtol=1e-5), (gloss, g["gloss"])
    check_table(gen.state_dict().items(), g["gen_names"], g["gen_stats"], rtol=1e-3, atol=1e-4, what="gen")
    sub = {k: v for k, v in m.state_dict().items() if ("pred_conv" in k or "running_mean" in k)}
    check_table(sub.items(), g["model_names"], g["model_stats"], rtol=1e-3, atol=1e-4, what="model")
    assert np.allclose(stats(m.backbone.conv1.weight), g["stem_stats"], rtol=0, atol=0)  # backbone untouched


@pytest.mark.parametrize("tag,avg_feat,context_aware", [("a", False, False), ("b", True, True)])
def test_gcn_context_trajectory_matches_reference(golden, torch_threads, tag, avg_feat, context_aware):
    """11 iterations of train_context_GMMN_GCNcontext.py:239-457 at 65x65, B=4: per-iteration classifier / generator /
    GCN-generator losses and the final generator, GCN generator and pred_conv weights.  (In this container the oracle
    reproduces the reference's 11-iteration trajectory bit for bit -- which needs the MMD's ops recorded in the
    reference's order, see mmd_loss; the tolerances leave room for a different CPU / thread count.)"""
    g = golden("gcn_traj.npz")
    seen = [c for c in range(21) if c not in (10, 14)]
    torch.manual_seed(1)
    m = zo.DeepLab(num_classes=21, pretrained=False)
    gen = zo.GMMNnetwork(300, 300, 256, 256)
    gcn = zo.GMMNnetwork_GCN(300, 300, 256, 256)
    torch.manual_seed(3)
    for layer in (gcn.gcn1, gcn.gcn2):
        nn.init.xavier_uniform_(layer.weight)
    assert np.array_equal(gcn.gcn1.weight.detach().numpy()[:8, :8], g[f"{tag}_gcn1_w0_corner"])
    params = [{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}]
    opt = torch.optim.SGD(params, momentum=0.9, weight_decay=5e-4, nesterov=False)
    opt_g = torch.optim.Adam(gen.parameters(), lr=2e-4)
    opt_c = torch.optim.Adam(gcn.parameters(), lr=2e-4)
    w = torch.ones(21)
    w[[10, 14]] = 100.0
    crit = zo.SegmentationLosses(weight=w).build_loss("ce")
    mmd = zo.GMMNLoss().build_loss()
    m.train(), gen.train(), gcn.train()
    torch.manual_seed(17)
    closs, gloss, gcnloss = [], [], []
    for it in range(11):
        b = zo.make_synthetic_batch(4, 65, seed=400 + it, with_label_emb=True)
        zo.apply_lr(opt, zo.poly_lr(0.007, it, 0, 11, 2))
        gl, gcl, cl = zo.gcn_context_step(m, gen, gcn, opt, opt_g, opt_c, crit, mmd, b["image"], b["label"], b["label_emb"],
                                          seen=seen, unseen=[10, 14], gcn_avg_feat=avg_feat, context_aware=context_aware)
        closs.append(cl), gloss.append(gl), gcnloss.append(gcl)
    assert np.allclose(closs, g[f"{tag}_closs"], rtol=1e-5), (closs, g[f"{tag}_closs"])
    assert np.allclose(gloss, g[f"{tag}_gloss"], rtol=1e-5), (gloss, g[f"{tag}_gloss"])
    assert np.allclose(gcnloss, g[f"{tag}_gcnloss"], rtol=1e-5), (gcnloss, g[f"{tag}_gcnloss"])
    check_table(gen.state_dict().items(), g[f"{tag}_gen_names"], g[f"{tag}_gen_stats"], rtol=1e-3, atol=1e-4, what="gen")
    check_table(gcn.state_dict().items(), g[f"{tag}_gcn_names"], g[f"{tag}_gcn_stats"], rtol=1e-3, atol=1e-4, what="gcn")
    sub = {k: v for k, v in m.state_dict().items() if "pred_conv" in k}
    check_table(sub.items(), g[f"{tag}_model_names"], g[f"{tag}_model_stats"], rtol=1e-3, atol=1e-4, what="model")
    assert np.allclose(gcn.gcn1.weight.detach().numpy()[:8, :8], g[f"{tag}_gcn1_w_corner"], rtol=1e-3, atol=1e-5)


def test_host_logic_matches_reference(golden):
    g = golden("misc.npz")
    lrs = [[zo.poly_lr(0.007, it, ep, 11, 3), 10 * zo.poly_lr(0.007, it, ep, 11, 3)] for ep in range(3) for it in range(11)]
    assert np.allclose(lrs, g["poly_lrs"], rtol=1e-12)
    cm = zo.confusion_matrix(g["cm_gt"], g["cm_pred"], 21)
    assert np.array_equal(cm, g["cm"])
    miou, by_class = zo.miou_from_confusion(cm)
    assert abs(miou - float(g["miou"])) < 1e-12 and np.allclose(by_class, g["miou_by_class"], equal_nan=True)
    seen = [c for c in range(21) if c not in (10, 14)]
    assert abs(np.nanmean(np.nan_to_num(by_class[seen])) - float(g["miou_seen"])) < 1e-12
    assert abs(np.nanmean(np.nan_to_num(by_class[[10, 14]])) - float(g["miou_unseen"])) < 1e-12
    assert abs(zo.pixel_accuracy(cm) - float(g["pix_acc"][0])) < 1e-12
    assert abs(zo.fw_iou(cm) - float(g["fwiou"][0])) < 1e-12
    assert np.array_equal(zo.nearest_index(129, 513), g["nearest_513_129"])
    assert np.array_equal(zo.nearest_index(17, 65), g["nearest_65_17"])
    assert list(zo.nearest_index(129, 513)[:4]) == [0, 3, 7, 11]


def test_cluster_graph_matches_reference_construct_adj_mat(golden):
    """oracle cluster_graph vs train_context_GMMN_GCNcontext.construct_adj_mat run in the build container
    (tests/golden/gcn_graph.npz): cluster numbering, dense adjacency, labels, seed embeddings and the seed-feature quirk"""
    g = golden("gcn_graph.npz")
    for k in range(int(g["n"])):
        adj, cmap, lbl, emb, feat = zo.cluster_graph(g[f"seg{k}"], g[f"emb{k}"], g[f"feat{k}"], avg_feat=(k % 2 == 1))
        assert np.array_equal(cmap, g[f"cmap{k}"]) and np.array_equal(lbl, g[f"lbl{k}"])
        assert (adj is not None) == bool(g[f"has_adj{k}"])
        if adj is not None:
            assert np.array_equal(adj, g[f"adj{k}"])
        assert np.array_equal(emb, g[f"emb_gcn{k}"]) and np.array_equal(feat, g[f"feat_gcn{k}"])
