#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (valeoai/ZS3 at /root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box).  The fixtures are
data: seeds, small input/output tensors and checksums.  Re-run:  python tools/make_goldens.py

Inputs come from zs3_oracle.make_synthetic_batch (a pure function of its seed), so fixtures store
seeds + input checksums rather than the inputs themselves.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ZS3_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, REF)


def _install_stubs():
    """Test-only stand-ins for packages the reference imports but this image lacks; none of them is on
    the numerical path that is being pinned (pygcn is only used by GMMNnetwork_GCN)."""
    if "pygcn" not in sys.modules:
        pygcn = types.ModuleType("pygcn")
        layers = types.ModuleType("pygcn.layers")

        class GraphConvolution(nn.Module):
            def __init__(self, fin, fout):
                super().__init__()
                self.weight = nn.Parameter(torch.zeros(fin, fout))
                self.bias = nn.Parameter(torch.zeros(fout))

            def forward(self, x, adj):
                return adj @ (x @ self.weight) + self.bias

        layers.GraphConvolution = GraphConvolution
        pygcn.layers = layers
        sys.modules["pygcn"], sys.modules["pygcn.layers"] = pygcn, layers
    for name in ("tensorboardX",):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.SummaryWriter = object
            sys.modules[name] = m
    try:
        import torchvision  # noqa: F401
    except Exception:
        tv = types.ModuleType("torchvision")
        tvu = types.ModuleType("torchvision.utils")
        tvu.make_grid = lambda *a, **k: None
        tvt = types.ModuleType("torchvision.transforms")
        tvt.Compose = object
        tv.utils, tv.transforms = tvu, tvt
        sys.modules.update({"torchvision": tv, "torchvision.utils": tvu, "torchvision.transforms": tvt})


_install_stubs()

import zs3_oracle as zo  # noqa: E402
from zs3.modeling.deeplab import DeepLab as RefDeepLab  # noqa: E402
from zs3.modeling.gmmn import GMMNnetwork as RefGMMN  # noqa: E402
from zs3.utils.loss import SegmentationLosses as RefSegLoss, GMMNLoss as RefGMMNLoss  # noqa: E402
from zs3.utils.lr_scheduler import LR_Scheduler as RefLR  # noqa: E402
from zs3.utils.metrics import Evaluator as RefEvaluator  # noqa: E402
from zs3.base_trainer import BaseTrainer as RefBaseTrainer  # noqa: E402


def stats(t):
    t = t.detach().double().reshape(-1)
    return np.array([t.sum().item(), t.abs().sum().item(), t[0].item(), t[-1].item()], dtype=np.float64)


def table(named):
    names, rows = [], []
    for k, v in named:
        names.append(k)
        rows.append(stats(v))
    return np.array(names), np.stack(rows)


def set_dropout(model, p=None):
    for m in model.modules():
        if isinstance(m, nn.Dropout) and p is not None:
            m.p = p


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------- G1: constructor init
def g_init():
    torch.manual_seed(1)
    m = RefDeepLab(num_classes=21, pretrained=False, sync_bn=False)
    n21, s21 = table(m.state_dict().items())
    lr1 = sum(p.numel() for p in m.get_1x_lr_params())
    lr10 = sum(p.numel() for p in m.get_10x_lr_params())
    torch.manual_seed(1)
    m60 = RefDeepLab(num_classes=60, pretrained=False, sync_bn=True, global_avg_pool_bn=False)
    n60, s60 = table(m60.state_dict().items())
    torch.manual_seed(1)
    g = RefGMMN(300, 300, 256, 256)
    ng, sg = table(g.state_dict().items())
    torch.manual_seed(3)
    g2 = RefGMMN(300, 300, 0, 256, semantic_reconstruction=True)
    ng2, sg2 = table(g2.state_dict().items())
    save("init.npz", names21=n21, stats21=s21, lr1=lr1, lr10=lr10, names60=n60, stats60=s60, names_g=ng, stats_g=sg,
         names_g2=ng2, stats_g2=sg2)


# --------------------------------------------------------------------------- G2: forward / backward
def g_forward():
    torch.manual_seed(1)
    m = RefDeepLab(num_classes=21, pretrained=False, sync_bn=False)
    batch = zo.make_synthetic_batch(2, 65, seed=7, with_label_emb=False)
    x, y = batch["image"], batch["label"]
    out = {"in_stats": stats(x), "label_stats": stats(y)}
    m.eval()
    with torch.no_grad():
        logits = m(x)
        feat = m.forward_before_class_prediction(x)
        top, low = m.backbone(x)
    out.update(eval_logits=logits.numpy(), eval_feat_stats=stats(feat), eval_feat_slice=feat[:, :8, ::4, ::4].numpy(),
               eval_top_stats=stats(top), eval_low_stats=stats(low), eval_argmax=logits.argmax(1).numpy().astype(np.uint8))
    # train mode, dropout disabled (BN batch statistics + running-stat update)
    m.train()
    set_dropout(m, 0.0)
    w = torch.ones(21)
    w[[10, 14]] = 100.0
    crit = RefSegLoss(weight=w, cuda=False).build_loss("ce")
    logits = m(x)
    loss = crit(logits, y)
    m.zero_grad()
    loss.backward()
    gn, gs = table((k, p.grad) for k, p in m.named_parameters())
    bn, bs = table((k, v) for k, v in m.state_dict().items() if "running" in k)
    out.update(train_logits=logits.detach().numpy(), train_loss=np.float64(loss.item()), grad_names=gn, grad_stats=gs,
               run_names=bn, run_stats=bs,
               grad_pred_w=m.decoder.pred_conv.weight.grad.numpy(), grad_stem_w=m.backbone.conv1.weight.grad.numpy()[:8])
    # other loss modes on the same logits
    lg = logits.detach()
    out["loss_focal"] = np.float64(RefSegLoss(weight=w).build_loss("focal")(lg, y).item())
    out["loss_ce_finetune"] = np.float64(RefSegLoss(weight=w).build_loss("ce_finetune")(lg, y).item())
    out["loss_ce_noweight"] = np.float64(RefSegLoss().build_loss("ce")(lg, y).item())
    # split forwards
    m.eval()
    with torch.no_grad():
        f4 = m.forward_before_last_conv_finetune(x)
        f8 = m.forward_class_last_conv_finetune(f4)
        lg2 = m.forward_class_prediction(f8, (65, 65))
    out.update(split_f4_stats=stats(f4), split_f8_stats=stats(f8), split_logits_stats=stats(lg2))
    save("deeplab_forward.npz", **out)


# --------------------------------------------------------------------------- G2b: BASELINE configs[0]
def g_config0():
    """BASELINE.json configs[0]: DeepLabv3+ ResNet-101, 21 classes, forward + CE loss on ONE random 3x129x129 tensor.
    (B = 1 raises in train() mode at the pooled-branch BN, aspp.py:87, so the reference's plumbing case is eval mode.)"""
    torch.manual_seed(1)
    m = RefDeepLab(num_classes=21, pretrained=False, sync_bn=False)
    m.eval()
    batch = zo.make_synthetic_batch(1, 129, seed=129, with_label_emb=False)
    x, y = batch["image"], batch["label"]
    with torch.no_grad():
        logits = m(x)
        loss = RefSegLoss(cuda=False).build_loss("ce")(logits, y)
    top2 = logits.topk(2, dim=1).values
    raised = False
    m.train()
    try:
        m(x)
    except ValueError:
        raised = True
    save("config0_129.npz", in_stats=stats(x), label_stats=stats(y), logits=logits.numpy(), loss=np.float64(loss.item()),
         argmax=logits.argmax(1).numpy().astype(np.uint8), margin=(top2[:, 0] - top2[:, 1]).numpy(),
         train_b1_raises=np.array(raised))


# --------------------------------------------------------------------------- G2c: the sizes the scripts and the bench really run
def g_sizes():
    """Round 6: the reference itself at 513 x 513 (the benchmark's resolution), at its default training crop
    312 x 312 (train_pascal.py:203-204) and with output_stride = 8 (resnet.py:72-74, aspp.py:49-50) -- pins the oracle where
    tests/test_gpu_parity_sizes.py uses it as the judge.  Same seeds / inputs as that test (B = 2, seed = size; train mode:
    seed = 1000 + size, residual-branch gains 0.1, dropout off, class weights 100 on classes 10 and 14)."""
    out = {}
    for size in (513, 312):
        torch.manual_seed(1)
        m = RefDeepLab(num_classes=21, pretrained=False, sync_bn=False).eval()
        x = zo.make_synthetic_batch(2, size, seed=size, with_label_emb=False)["image"]
        with torch.no_grad():
            logits = m(x)
        out.update({f"eval{size}_in_stats": stats(x), f"eval{size}_logits_sub": logits[:, :, ::8, ::8].numpy().copy(),
                    f"eval{size}_logits_stats": stats(logits), f"eval{size}_argmax": logits.argmax(1).numpy().astype(np.uint8)})
        print(f"  eval {size}: done")
    # train mode at the default crop
    torch.manual_seed(1)
    m = RefDeepLab(num_classes=21, pretrained=False, sync_bn=False)
    for name, mod in m.named_modules():
        if name.endswith("bn3"):
            mod.weight.data.fill_(0.1)
    set_dropout(m, 0.0)
    m.train()
    b = zo.make_synthetic_batch(2, 312, seed=1312, with_label_emb=False)
    w = torch.ones(21)
    w[[10, 14]] = 100.0
    logits = m(b["image"])
    loss = RefSegLoss(weight=w, cuda=False).build_loss("ce")(logits, b["label"])
    loss.backward()
    gn, gs = table((k, p.grad) for k, p in m.named_parameters())
    rn, rs = table((k, v) for k, v in m.state_dict().items() if "running" in k)
    out.update(train312_logits_sub=logits.detach()[:, :, ::8, ::8].numpy().copy(), train312_logits_stats=stats(logits),
               train312_loss=np.float64(loss.item()), train312_grad_names=gn, train312_grad_stats=gs, train312_run_names=rn,
               train312_run_stats=rs, train312_grad_pred_w=m.decoder.pred_conv.weight.grad.numpy().copy())
    print("  train 312: done")
    # output_stride = 8
    torch.manual_seed(1)
    m = RefDeepLab(output_stride=8, num_classes=21, pretrained=False, sync_bn=False)
    for name, mod in m.named_modules():
        if name.endswith("bn3"):
            mod.weight.data.fill_(0.1)
    m.eval()
    x = zo.make_synthetic_batch(2, 65, seed=8, with_label_emb=False)["image"]
    with torch.no_grad():
        logits = m(x)
        top, low = m.backbone(x)
    out.update(os8_logits=logits.numpy().copy(), os8_top_stats=stats(top), os8_low_stats=stats(low),
               os8_top_shape=np.array(top.shape), os8_argmax=logits.argmax(1).numpy().astype(np.uint8))
    save("sizes.npz", **out)


# --------------------------------------------------------------------------- G3: supervised trajectory
class _Writer:
    def __init__(self):
        self.scalars = {}

    def add_scalar(self, tag, value, step):
        self.scalars.setdefault(tag, []).append((step, float(value)))


class _Summary:
    def visualize_image(self, *a, **k):
        pass


class _Module(nn.Module):
    """CPU passthrough that supplies `.module` like nn.DataParallel does on GPU."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def _loader(n_iter, batch, size, seed0, with_emb):
    out = []
    for it in range(n_iter):
        b = zo.make_synthetic_batch(batch, size, seed=seed0 + it, with_label_emb=with_emb)
        out.append(b)
    return out


def g_supervised():
    """base_trainer.py:5-25 driven for 11 iterations (its `num_img_tr // 10` needs >= 10 batches) at 65x65, B=2,
    small LR so that the comparison pins the update rule (poly LR, 1x/10x groups, momentum, weight decay)
    rather than chaotic amplification through tiny-batch BN."""
    torch.manual_seed(1)
    m = RefDeepLab(num_classes=21, pretrained=False, sync_bn=False)
    init = {k: v.clone() for k, v in m.state_dict().items()}
    params = [{"params": m.get_1x_lr_params(), "lr": 1e-5}, {"params": m.get_10x_lr_params(), "lr": 1e-4}]
    opt = torch.optim.SGD(params, momentum=0.9, weight_decay=5e-4, nesterov=False)
    tr = RefBaseTrainer.__new__(RefBaseTrainer)
    tr.model = _Module(m)
    tr.train_loader = _loader(11, 2, 65, 100, False)
    tr.args = types.SimpleNamespace(cuda=False, batch_size=2, dataset="pascal", no_val=False)
    tr.scheduler = RefLR("poly", 1e-5, 2, 11)
    tr.optimizer = opt
    tr.criterion = RefSegLoss(cuda=False).build_loss("ce")
    tr.best_pred = 0.0
    tr.writer = _Writer()
    tr.summary = _Summary()
    torch.manual_seed(11)  # dropout stream of the trajectory
    tr.training(0)
    losses = np.array([v for _, v in tr.writer.scalars["train/total_loss_iter"]])
    keys = [k for k, v in m.state_dict().items() if v.dim() == 4 or k.endswith("pred_conv.bias") or k.endswith("bn1.weight")]
    n, s = table((k, m.state_dict()[k] - init[k]) for k in keys)
    save("supervised_traj.npz", losses=losses, delta_names=n, delta_stats=s,
         final_lr=np.array([g["lr"] for g in opt.param_groups]),
         nbt=np.int64(m.state_dict()["backbone.bn1.num_batches_tracked"].item()))


# --------------------------------------------------------------------------- G4/G5: MMD + GMMN MLP
def g_mmd():
    crit = RefGMMNLoss(sigma=[2, 5, 10, 20, 40, 80]).build_loss()
    out = {}
    cases = {"rand128": (128, 256, 1.0, 0.3), "far128": (128, 256, 0.2, 0.0), "small4": (4, 8, 1.0, 0.5),
             "near128": (128, 256, 0.5, 0.98)}
    for name, (n, d, scale, mix) in cases.items():
        g = torch.Generator().manual_seed(len(name) * 17 + n)
        real = torch.randn(n, d, generator=g) * scale
        gen = (mix * real + (1 - mix) * torch.randn(n, d, generator=g) * scale).requires_grad_(True)
        loss = crit(gen, real)
        loss.backward()
        out[f"{name}_gen"], out[f"{name}_real"] = gen.detach().numpy(), real.numpy()
        out[f"{name}_loss"], out[f"{name}_grad"] = np.float64(loss.item()), gen.grad.numpy()
    real = torch.randn(128, 256, generator=torch.Generator().manual_seed(5))
    out["identical_loss"] = np.float64(crit(real.clone(), real).item())
    save("mmd.npz", **out)


def g_gmmn_mlp():
    torch.manual_seed(1)
    net = RefGMMN(300, 300, 256, 256)
    net.eval()
    g = torch.Generator().manual_seed(21)
    emb = torch.randn(37, 300, generator=g).requires_grad_(True)
    z = torch.rand(37, 300, generator=g)
    y = net(emb, z)
    up = torch.randn(37, 256, generator=g)
    (y * up).sum().backward()
    gn, gs = table((k, p.grad) for k, p in net.named_parameters())
    save("gmmn_mlp.npz", out=y.detach().numpy(), up=up.numpy(), grad_names=gn, grad_stats=gs, grad_emb_stats=stats(emb.grad),
         grad_w2=net.model[3].weight.grad.numpy()[:16, :16])


# --------------------------------------------------------------------------- G6: GMMN trajectory
def g_gmmn_traj():
    import zs3.train_pascal_GMMN as T

    seen = [c for c in range(21) if c not in (10, 14)]
    unseen = [10, 14]
    args = types.SimpleNamespace(cuda=False, batch_size=2, dataset="pascal", no_val=False, feature_dim=256, embed_dim=300,
                                 noise_dim=300, batch_size_generator=128, unseen_classes_idx_metric=unseen,
                                 seen_classes_idx_metric=seen, real_seen_features=True)
    torch.manual_seed(1)
    m = RefDeepLab(num_classes=21, pretrained=False, sync_bn=False)
    gen = RefGMMN(300, 300, 256, 256)
    params = [{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}]
    tr = T.Trainer.__new__(T.Trainer)
    tr.args = args
    tr.model = _Module(m)
    tr.generator = gen
    tr.optimizer = torch.optim.SGD(params, momentum=0.9, weight_decay=5e-4, nesterov=False)
    tr.optimizer_generator = torch.optim.Adam(gen.parameters(), lr=2e-4)
    w = torch.ones(21)
    w[unseen] = 100.0
    tr.criterion = RefSegLoss(weight=w, cuda=False).build_loss("ce")
    tr.criterion_generator = RefGMMNLoss(sigma=[2, 5, 10, 20, 40, 80]).build_loss()
    tr.scheduler = RefLR("poly", 0.007, 2, 11)
    tr.best_pred = 0.0
    tr.writer = _Writer()
    tr.summary = _Summary()
    # batch 4 so that image index 3 holds an unseen class (generated-feature branch) and 0..2 train the generator
    tr.train_loader = _loader(11, 4, 65, 200, True)
    torch.manual_seed(13)  # CPU stream: dropout masks, z noise, MMD sample indices
    tr.training(0, args)
    closs = np.array([v for _, v in tr.writer.scalars["train/total_loss_iter"]])
    gloss = np.array([v for _, v in tr.writer.scalars["train/generator_loss"]])
    gn, gs = table(gen.state_dict().items())
    mn, ms = table((k, v) for k, v in m.state_dict().items() if ("pred_conv" in k or "running_mean" in k))
    untouched = stats(m.backbone.conv1.weight)
    save("gmmn_traj.npz", closs=closs, gloss=gloss, gen_names=gn, gen_stats=gs, model_names=mn, model_stats=ms,
         stem_stats=untouched)


# --------------------------------------------------------------------------- G7: misc host logic
def g_misc():
    sch = RefLR("poly", 0.007, 3, 11)
    opt = torch.optim.SGD([{"params": [nn.Parameter(torch.zeros(1))]}, {"params": [nn.Parameter(torch.zeros(1))]}], lr=0.1)
    lrs = []
    for ep in range(3):
        for it in range(11):
            sch(opt, it, ep, 0.0)
            lrs.append([g["lr"] for g in opt.param_groups])
    rng = np.random.RandomState(3)
    gt = rng.randint(0, 21, size=(4, 33, 33))
    gt[0, :3] = 255
    pred = np.where(rng.rand(4, 33, 33) < 0.7, np.minimum(gt, 20), rng.randint(0, 21, size=(4, 33, 33)))
    ev = RefEvaluator(21, [c for c in range(21) if c not in (10, 14)], [10, 14])
    ev.add_batch(gt, pred)
    miou, by_class, miou_seen, miou_unseen = ev.Mean_Intersection_over_Union()
    acc, acc_seen, acc_unseen = ev.Pixel_Accuracy()
    acc_cls, acc_by_class, acc_cls_seen, acc_cls_unseen = ev.Pixel_Accuracy_Class()
    fw, fw_seen, fw_unseen = ev.Frequency_Weighted_Intersection_over_Union()
    ev_plain = RefEvaluator(21)            # no seen/unseen split: scalar / 2-tuple return forms
    ev_plain.add_batch(gt, pred)
    near = torch.nn.functional.interpolate(torch.arange(513.0).view(1, 1, 1, 513), size=(1, 129), mode="nearest").view(-1)
    near65 = torch.nn.functional.interpolate(torch.arange(65.0).view(1, 1, 1, 65), size=(1, 17), mode="nearest").view(-1)
    save("misc.npz", poly_lrs=np.array(lrs), cm_gt=gt.astype(np.uint8), cm_pred=pred.astype(np.uint8),
         cm=ev.confusion_matrix, miou=np.float64(miou), miou_by_class=by_class, miou_seen=np.float64(miou_seen),
         miou_unseen=np.float64(miou_unseen),
         pix_acc=np.array([acc, acc_seen, acc_unseen]), pix_acc_class=np.array([acc_cls, acc_cls_seen, acc_cls_unseen]),
         pix_acc_by_class=acc_by_class, fwiou=np.array([fw, fw_seen, fw_unseen]),
         plain=np.array([ev_plain.Pixel_Accuracy(), ev_plain.Pixel_Accuracy_Class()[0],
                         ev_plain.Mean_Intersection_over_Union()[0], ev_plain.Frequency_Weighted_Intersection_over_Union()]),
         nearest_513_129=near.numpy().astype(np.int64),
         nearest_65_17=near65.numpy().astype(np.int64))


def g_seen_unseen():
    """Evaluator_seen_unseen.label_accuracy_score (metrics.py:88-196; eval_pascal.py:83) on four seeded label / prediction
    maps with an ignore band and one class missing from an image: overall / seen / unseen tuples and the per-class list."""
    import warnings
    from zs3.utils.metrics import Evaluator_seen_unseen as RefESU
    rng = np.random.RandomState(5)
    gt = rng.randint(0, 21, size=(4, 33, 33))
    gt[:, :3] = 255
    gt[1][gt[1] == 5] = 7
    pred = np.where(rng.rand(4, 33, 33) < 0.6, np.minimum(gt, 20), rng.randint(0, 21, size=(4, 33, 33)))

    def flat(x, out):
        if isinstance(x, (tuple, list)):
            for y in x:
                flat(y, out)
        else:
            out.append(float(x))
        return out
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        split = RefESU(21, [10, 14]).label_accuracy_score(list(gt), list(pred), by_class=True)
        plain = RefESU(21, None).label_accuracy_score(list(gt), list(pred))
    save("seen_unseen.npz", gt=gt.astype(np.uint8), pred=pred.astype(np.uint8), split=np.array(flat(split, [])),
         plain=np.array(flat(plain, [])))


# --------------------------------------------------------------------------- G8: GCN-context cluster graph (8f N3)
def _segmaps():
    """label maps with blobs, thin diagonal structures (8-connectivity matters), single-pixel clusters, one-label maps"""
    rng = np.random.RandomState(7)
    maps = []
    m = np.zeros((17, 23), dtype=np.int64)
    m[2:9, 3:12] = 4; m[5:14, 10:20] = 7; m[12:, :6] = 4
    for k in range(10):
        m[k, 22 - k] = 9                       # anti-diagonal line: one cluster only under 8-connectivity
    m[15, 15] = 3; m[0, 0] = 3
    maps.append(m)
    coarse = rng.randint(0, 5, size=(5, 6))
    maps.append(np.kron(coarse, np.ones((7, 6), dtype=np.int64)))   # 35 x 36 block map
    maps.append(rng.randint(0, 3, size=(12, 13)))                    # salt-and-pepper: many tiny clusters
    maps.append(np.full((9, 9), 2, dtype=np.int64))                  # a single cluster (adj_mat is None)
    big = np.zeros((33, 33), dtype=np.int64)
    yy, xx = np.mgrid[:33, :33]
    big[(yy - 10) ** 2 + (xx - 12) ** 2 < 49] = 5
    big[(yy - 22) ** 2 + (xx - 20) ** 2 < 64] = 11
    big[(yy + xx) % 17 == 0] = 2
    big[30:, :] = 255
    maps.append(big)
    return maps


def g_gcn():
    from zs3.train_context_GMMN_GCNcontext import construct_adj_mat
    out = {}
    rng = np.random.RandomState(11)
    for k, seg in enumerate(_segmaps()):
        emb = rng.randn(6, *seg.shape).astype(np.float32)
        feat = rng.randn(5, *seg.shape).astype(np.float32)
        adj, c2p, c2l, emb_gcn, feat_gcn = construct_adj_mat(seg, emb, feat, avg_feat=(k % 2 == 1))
        ncl = len(c2l)
        cmap = np.full(seg.shape, -1, dtype=np.int64)
        for c, pix in c2p.items():
            for (i, j) in pix:
                cmap[i, j] = c
        dense = np.zeros((ncl, ncl), dtype=np.float32) if adj is None else adj.to_dense().numpy()
        out.update({f"seg{k}": seg, f"emb{k}": emb, f"feat{k}": feat, f"cmap{k}": cmap, f"adj{k}": dense,
                    f"has_adj{k}": np.array(adj is not None), f"lbl{k}": np.array(c2l, dtype=np.int64),
                    f"emb_gcn{k}": np.asarray(emb_gcn), f"feat_gcn{k}": np.asarray(feat_gcn)})
    out["n"] = np.array(len(_segmaps()))
    save("gcn_graph.npz", **out)


# --------------------------------------------------------------------------- G9: GCN-context trajectory (8f N3)
def g_gcn_traj():
    """train_context_GMMN_GCNcontext.py Trainer.training driven on CPU.  The loop calls .cuda() unconditionally on the
    cluster tensors (:400-412, :441-449); for this CPU run Tensor.cuda is replaced by the identity while it executes."""
    import zs3.train_context_GMMN_GCNcontext as T
    from zs3.modeling.gmmn import GMMNnetwork_GCN as RefGCN

    seen = [c for c in range(21) if c not in (10, 14)]
    unseen = [10, 14]
    out = {}
    for tag, avg_feat, context_aware in (("a", False, False), ("b", True, True)):
        args = types.SimpleNamespace(cuda=False, batch_size=2, dataset="pascal", no_val=False, feature_dim=256, embed_dim=300,
                                     noise_dim=300, batch_size_generator=128, unseen_classes_idx_metric=unseen,
                                     seen_classes_idx_metric=seen, real_seen_features=True, context_aware=context_aware,
                                     GCN_weight=0.1, GCN_avg_feat=avg_feat, semantic_reconstruction=False)
        torch.manual_seed(1)
        m = RefDeepLab(num_classes=21, pretrained=False, sync_bn=False)
        gen = RefGMMN(300, 300, 256, 256)
        gcn = RefGCN(300, 300, 256, 256)
        # pygcn's own reset_parameters (not reproduced by the import stub) consumes RNG before the reference's xavier
        # init: re-draw the weights under their own seed so that a checker can start from the same generator
        torch.manual_seed(3)
        for layer in (gcn.gcn1, gcn.gcn2):
            torch.nn.init.xavier_uniform_(layer.weight)
        params = [{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}]
        tr = T.Trainer.__new__(T.Trainer)
        tr.args = args
        tr.model = _Module(m)
        tr.generator, tr.generator_GCN = gen, gcn
        tr.optimizer = torch.optim.SGD(params, momentum=0.9, weight_decay=5e-4, nesterov=False)
        tr.optimizer_generator = torch.optim.Adam(gen.parameters(), lr=2e-4)
        tr.optimizer_generator_GCN = torch.optim.Adam(gcn.parameters(), lr=2e-4)
        w = torch.ones(21)
        w[unseen] = 100.0
        tr.criterion = RefSegLoss(weight=w, cuda=False).build_loss("ce")
        tr.criterion_generator = RefGMMNLoss(sigma=[2, 5, 10, 20, 40, 80]).build_loss()
        tr.scheduler = RefLR("poly", 0.007, 2, 11)
        tr.best_pred = 0.0
        tr.writer = _Writer()
        tr.summary = _Summary()
        tr.train_loader = _loader(11, 4, 65, 400, True)
        init = {k: v.clone() for k, v in gcn.state_dict().items()}
        torch.manual_seed(17)
        real_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            tr.training(0, args)
        finally:
            torch.Tensor.cuda = real_cuda
        sc = tr.writer.scalars
        gn, gs = table(gen.state_dict().items())
        cn, cs = table(gcn.state_dict().items())
        mn, ms = table((k, v) for k, v in m.state_dict().items() if "pred_conv" in k)
        out.update({f"{tag}_closs": np.array([v for _, v in sc["train/total_loss_iter"]]),
                    f"{tag}_gloss": np.array([v for _, v in sc["train/generator_loss"]]),
                    f"{tag}_gcnloss": np.array([v for _, v in sc["train/generator_GCN_loss"]]),
                    f"{tag}_gen_names": gn, f"{tag}_gen_stats": gs, f"{tag}_gcn_names": cn, f"{tag}_gcn_stats": cs,
                    f"{tag}_model_names": mn, f"{tag}_model_stats": ms,
                    f"{tag}_gcn1_w_corner": gcn.gcn1.weight.detach().numpy()[:8, :8].copy(),
                    f"{tag}_gcn1_w0_corner": init["gcn1.weight"].numpy()[:8, :8].copy()})
    save("gcn_traj.npz", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["init", "forward", "config0", "sizes", "supervised", "mmd", "gmmn_mlp", "gmmn_traj", "misc", "seen_unseen", "gcn", "gcn_traj"]
    for w in which:
        {"seen_unseen": g_seen_unseen, "sizes": g_sizes, "init": g_init, "forward": g_forward, "config0": g_config0, "supervised": g_supervised, "mmd": g_mmd, "gmmn_mlp": g_gmmn_mlp,
         "gmmn_traj": g_gmmn_traj, "misc": g_misc, "gcn": g_gcn, "gcn_traj": g_gcn_traj}[w]()
