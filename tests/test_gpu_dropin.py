"""GPU tests of the drop-in SURFACE (BASELINE configs[3]): the reference's scripts keep
their own loops and only see zs3_amd through the module paths, constructors and trainer entry points they already use.

* the `sys.modules` alias of INTEGRATION.md section 2, driven by a loop body written the way
  zs3/train_pascal_GMMN.py:139-268 is written (eager `GMMNnetwork` over ALL pixels of a class, `GMMNLoss` through
  autograd, `torch.optim.Adam`/`SGD`, boolean masks, `.item()`), against the oracle and the reference's golden trajectory;
* `BaseTrainer.training(epoch)` (zs3/base_trainer.py:5-57) with injected stubs, against the oracle and
  tests/golden/supervised_traj.npz;
* `GMMNTrainer.training(epoch, args)`;
* `GMMNStep` at 60 classes (train_context_GMMN.py, datasets/context.py:22) and `GMMNStep(table=...)` against the oracle.

Only fixtures and the oracle are used: nothing here reads /root/reference."""
import sys
import types

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


def _groups(mod, lr):
    return [{"params": mod.get_1x_lr_params(), "lr": lr}, {"params": mod.get_10x_lr_params(), "lr": lr * 10}]


def _tame(model):
    for name, mod in model.named_modules():
        if name.endswith("bn3"):
            mod.weight.data.fill_(0.1)


def _no_dropout(*mods):
    for m in mods:
        for sub in m.modules():
            if isinstance(sub, nn.Dropout):
                sub.p = 0.0


@pytest.fixture()
def zs3_alias():
    """INTEGRATION.md section 2, second form: alias the hot-path modules under the reference's names.  On the GPU box the
    reference package does not exist, so empty parent packages stand in for `zs3`, `zs3.modeling`, `zs3.utils`."""
    import zs3_amd.base_trainer
    import zs3_amd.modeling.deeplab
    import zs3_amd.modeling.gmmn
    import zs3_amd.modeling.sync_batchnorm.batchnorm as sbn
    import zs3_amd.modeling.sync_batchnorm.replicate as rep
    import zs3_amd.utils.loss
    saved = {k: v for k, v in sys.modules.items() if k == "zs3" or k.startswith("zs3.")}
    for k in saved:
        del sys.modules[k]
    for pkg in ("zs3", "zs3.modeling", "zs3.utils", "zs3.modeling.sync_batchnorm"):
        mod = types.ModuleType(pkg)
        mod.__path__ = []
        sys.modules[pkg] = mod
    sys.modules.update({
        "zs3.modeling.deeplab": zs3_amd.modeling.deeplab, "zs3.modeling.gmmn": zs3_amd.modeling.gmmn,
        "zs3.utils.loss": zs3_amd.utils.loss, "zs3.base_trainer": zs3_amd.base_trainer,
        "zs3.modeling.sync_batchnorm.replicate": rep, "zs3.modeling.sync_batchnorm.batchnorm": sbn})
    yield
    for k in [k for k in sys.modules if k == "zs3" or k.startswith("zs3.")]:
        del sys.modules[k]
    sys.modules.update(saved)


class _Passthrough(nn.Module):
    """what nn.DataParallel(model, device_ids=[0]) is with one process per GPU: `.module` + a plain forward"""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def _script_style_gmmn_iteration(model, generator, optimizer, optimizer_generator, criterion, criterion_generator, sample,
                                 args):
    """One iteration the way the reference's script does it (train_pascal_GMMN.py:139-268), using nothing but the
    drop-in surface: `model.module.forward_before_class_prediction`, eager generator calls on every pixel of a class,
    autograd through GMMNLoss, the caller's torch optimizers, `.item()` per class."""
    image, target, embedding = sample["image"].cuda(), sample["label"].cuda(), sample["label_emb"].cuda()
    with torch.no_grad():
        real_features = model.module.forward_before_class_prediction(image)
    fake_features = torch.zeros(real_features.shape, device=real_features.device)
    fh, fw = real_features.shape[2:]
    generator_loss_batch = 0.0
    seen = [float(c) for c in args.seen_classes_idx_metric]
    unseen = [float(c) for c in args.unseen_classes_idx_metric]
    for n in range(image.shape[0]):
        feats = real_features[n].permute(1, 2, 0).reshape(-1, args.feature_dim)
        labels = nn.functional.interpolate(target[n].view(1, 1, *target.shape[1:]), size=(fh, fw), mode="nearest").view(-1)
        emb = nn.functional.interpolate(embedding[n].unsqueeze(0), size=(fh, fw), mode="nearest")[0]
        emb = emb.permute(1, 2, 0).reshape(-1, args.embed_dim)
        generated = torch.zeros_like(feats)
        present = labels.unique()
        image_has_unseen = bool(sum(float(c) in unseen for c in present))
        running = 0.0
        for c in present:
            if float(c) == 255:
                continue
            optimizer_generator.zero_grad()
            where = labels == c
            count = int(where.sum().item())
            noise = torch.rand((count, args.noise_dim)).cuda()
            made = generator(emb[where], noise.float())
            if float(c) in seen and not image_has_unseen:
                pick = torch.randint(low=0, high=count, size=(args.batch_size_generator,)).cuda()
                g_loss = criterion_generator(made[pick], feats[where][pick])
                running += g_loss.item()
                g_loss.backward()
                optimizer_generator.step()
            generated[where] = made.detach()
        generator_loss_batch += running / len(present)
        keep_real = args.real_seen_features and not image_has_unseen
        fake_features[n] = (feats if keep_real else generated).reshape(fh, fw, args.feature_dim).permute(2, 0, 1)
    optimizer.zero_grad()
    output = model.module.forward_class_prediction(fake_features.detach(), image.shape[2:])
    loss = criterion(output, target)
    loss.backward()
    optimizer.step()
    return generator_loss_batch, loss.item()


def test_reference_style_gmmn_loop_through_module_alias(dev, golden, zs3_alias):
    """The reference's own loop shape on the aliased modules.  (1) dropout off, trained-ResNet-like BN gains, CPU noise
    stream: two iterations agree with the oracle's restatement of the same loop; (2) default constructor init and live
    dropout, i.e. the configuration tests/golden/gmmn_traj.npz was recorded from the reference in: the first iterations'
    losses agree with the reference's up to what different dropout masks do (device RNG vs the reference's CPU stream)."""
    import zs3_oracle as zo
    from zs3.modeling.deeplab import DeepLab            # the aliased (zs3_amd) modules, under the reference's names
    from zs3.modeling.gmmn import GMMNnetwork
    from zs3.modeling.sync_batchnorm.replicate import patch_replication_callback
    from zs3.utils.loss import GMMNLoss, SegmentationLosses
    import zs3_amd.modeling.deeplab
    from zs3_amd import functional as Fz
    assert DeepLab is zs3_amd.modeling.deeplab.DeepLab
    with pytest.raises(AssertionError):
        patch_replication_callback(nn.Linear(1, 1))      # replicate.py:58 asserts a DataParallel
    args = types.SimpleNamespace(seen_classes_idx_metric=[c for c in range(21) if c not in (10, 14)],
                                 unseen_classes_idx_metric=[10, 14], noise_dim=300, embed_dim=300, feature_dim=256,
                                 batch_size_generator=128, real_seen_features=True)
    w = torch.ones(21)
    w[[10, 14]] = 100.0

    # ---- (1) against the oracle
    torch.manual_seed(1)
    net = DeepLab(num_classes=21, pretrained=False)
    _tame(net)
    ref = zo.DeepLab(num_classes=21, pretrained=False)
    ref.load_state_dict(net.state_dict())
    torch.manual_seed(2)
    gen = GMMNnetwork(300, 300, 256, 256)
    gen_r = zo.GMMNnetwork(300, 300, 256, 256)
    gen_r.load_state_dict(gen.state_dict())
    _no_dropout(net, ref, gen, gen_r)
    model = _Passthrough(net.cuda()).train()
    gen = gen.cuda().train()
    ref.train()
    gen_r.train()
    opt = torch.optim.SGD(_groups(model.module, 0.007), momentum=0.9, weight_decay=5e-4, nesterov=False)
    opt_g = torch.optim.Adam(gen.parameters(), lr=2e-4)
    opt_r = torch.optim.SGD(_groups(ref, 0.007), momentum=0.9, weight_decay=5e-4, nesterov=False)
    opt_gr = torch.optim.Adam(gen_r.parameters(), lr=2e-4)
    crit = SegmentationLosses(weight=w.cuda(), cuda=True).build_loss(mode="ce")
    crit_g = GMMNLoss(sigma=[2, 5, 10, 20, 40, 80], cuda=True).build_loss()
    for it in range(2):
        b = zo.make_synthetic_batch(4, 65, seed=200 + it, with_label_emb=True)
        torch.manual_seed(13 + it)
        gl_r, cl_r = zo.gmmn_step(ref, gen_r, opt_r, opt_gr, zo.SegmentationLosses(weight=w).build_loss("ce"),
                                  zo.GMMNLoss().build_loss(), b["image"], b["label"], b["label_emb"],
                                  seen=args.seen_classes_idx_metric, unseen=[10, 14])
        torch.manual_seed(13 + it)
        gl, cl = _script_style_gmmn_iteration(model, gen, opt, opt_g, crit, crit_g, b, args)
        assert abs(gl - gl_r) < 2e-3 * abs(gl_r), (it, gl, gl_r)
        assert abs(cl - cl_r) < 1e-3 * abs(cl_r), (it, cl, cl_r)
    for (k, p), (_, pr) in zip(gen.named_parameters(), gen_r.named_parameters()):
        assert rel(p, pr) < 2e-2, k
        assert ((p.detach().cpu() - pr.detach()).abs().mean() / pr.detach().abs().mean()).item() < 2e-3, k
    assert rel(model.module.decoder.pred_conv.weight, ref.decoder.pred_conv.weight) < 2e-3

    # ---- (2) against the reference's recorded trajectory (default init, dropout live)
    g = golden("gmmn_traj.npz")
    torch.manual_seed(1)
    net = DeepLab(num_classes=21, pretrained=False)
    gen = GMMNnetwork(300, 300, 256, 256)
    model = _Passthrough(net.cuda()).train()
    gen = gen.cuda().train()
    opt = torch.optim.SGD(_groups(model.module, 0.007), momentum=0.9, weight_decay=5e-4, nesterov=False)
    opt_g = torch.optim.Adam(gen.parameters(), lr=2e-4)
    torch.manual_seed(13)
    Fz.manual_seed(5)      # the device dropout stream: fixed, so the outcome does not depend on which tests ran before
    closs, gloss = [], []
    for it in range(3):
        b = zo.make_synthetic_batch(4, 65, seed=200 + it, with_label_emb=True)
        zo.apply_lr(opt, zo.poly_lr(0.007, it, 0, 11, 2))
        gl, cl = _script_style_gmmn_iteration(model, gen, opt, opt_g, crit, crit_g, b, args)
        closs.append(cl)
        gloss.append(gl)
    print("gmmn alias trajectory:", closs, g["closs"][:3], gloss, g["gloss"][:3])
    # different dropout masks (decoder p=0.5/0.1, generator p=0.5): the reference's own spread between iterations
    # is 0.68-0.83 (classifier) and 3.8-4.3 (generator); same weights, same batches, same noise stream otherwise.  With
    # other mask draws single classifier iterations have landed 18 % off the recorded value, hence 0.25 there.
    assert np.allclose(closs, g["closs"][:3], rtol=0.25), (closs, g["closs"][:3])
    assert np.allclose(gloss, g["gloss"][:3], rtol=0.15), (gloss, g["gloss"][:3])


# ------------------------------------------------------------------------------------------- BaseTrainer.training
class _Recorder:
    def __init__(self):
        self.scalars, self.images, self.checkpoints = [], 0, []

    def add_scalar(self, tag, value, step):
        self.scalars.append((tag, float(value), int(step)))

    def visualize_image(self, writer, dataset, image, target, output, step):
        assert output.shape[0] == image.shape[0] and output.shape[2:] == image.shape[2:]
        self.images += 1

    def save_checkpoint(self, state, is_best):
        self.checkpoints.append((sorted(state), is_best))


def _supervised_loader(n, batch, size, seed0):
    import zs3_oracle as zo
    out = []
    for it in range(n):
        b = zo.make_synthetic_batch(batch, size, seed=seed0 + it, with_label_emb=False)
        out.append({"image": b["image"], "label": b["label"]})
    return out


def _make_trainer(model, optimizer, criterion, loader, scheduler, cuda):
    from zs3_amd.base_trainer import BaseTrainer

    class Trainer(BaseTrainer):      # the scripts' Trainer classes subclass BaseTrainer and set these attributes
        pass

    t = Trainer()
    rec = _Recorder()
    t.model, t.optimizer, t.criterion, t.train_loader, t.scheduler = model, optimizer, criterion, loader, scheduler
    t.args = types.SimpleNamespace(cuda=cuda, batch_size=2, dataset="pascal", no_val=True)
    t.best_pred, t.writer, t.summary, t.saver = 0.0, rec, rec, rec
    return t, rec


def test_base_trainer_training_epoch(dev, golden):
    """`BaseTrainer.training(epoch)` (base_trainer.py:5-57) end to end with injected stubs: skip of single-sample
    batches, LR schedule before each step, per-iteration scalar, ten image dumps, per-epoch checkpoint with no_val.
    (1) dropout off: the logged losses follow the oracle trainer run on the same batches; (2) default init with live
    dropout over the 11 batches of tests/golden/supervised_traj.npz: losses agree with the reference's recorded ones up
    to the effect of different dropout masks."""
    import zs3_oracle as zo
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.optim import SGD
    from zs3_amd.utils.loss import SegmentationLosses
    from zs3_amd.utils.lr_scheduler import LR_Scheduler
    g = golden("supervised_traj.npz")
    loader = _supervised_loader(11, 2, 65, 100)
    loader.insert(5, {"image": loader[0]["image"][:1], "label": loader[0]["label"][:1]})   # must be skipped (:11)

    # ---- (1) oracle, dropout off, tamed
    torch.manual_seed(1)
    net = DeepLab(num_classes=21, pretrained=False)
    _tame(net)
    ref = zo.DeepLab(num_classes=21, pretrained=False)
    ref.load_state_dict(net.state_dict())
    _no_dropout(net, ref)
    model = _Passthrough(net.to(dev))
    opt = SGD(_groups(net, 1e-3), momentum=0.9, weight_decay=5e-4, nesterov=False)
    trainer, rec = _make_trainer(model, opt, SegmentationLosses(cuda=True).build_loss("ce"), loader,
                                 LR_Scheduler("poly", 1e-3, 2, len(loader), verbose=False), cuda=True)
    trainer.training(0)
    ref.train()
    opt_r = torch.optim.SGD(_groups(ref, 1e-3), momentum=0.9, weight_decay=5e-4, nesterov=False)
    crit_r = zo.SegmentationLosses().build_loss("ce")
    want = []
    for i, sample in enumerate(loader):
        if len(sample["image"]) <= 1:
            continue
        zo.apply_lr(opt_r, zo.poly_lr(1e-3, i, 0, len(loader), 2))
        want.append(zo.supervised_step(ref, opt_r, crit_r, sample["image"], sample["label"])[0])
    got = [v for tag, v, _ in rec.scalars if tag == "train/total_loss_iter"]
    steps = [s for tag, _, s in rec.scalars if tag == "train/total_loss_iter"]
    assert len(got) == 11 and steps == [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11]          # index 5 skipped, step = index
    print("base trainer vs oracle:", got, want)
    # two fp32 runs of a network this deep drift apart under SGD: the first step pins the arithmetic
    # (no update has happened yet), the next ones the update, the rest the LOOP -- rtol 1e-1 from step 3 on says "same schedule,
    # same skipping, same accumulation", it is not an arithmetic bound (those are test_supervised_step_vs_oracle's UPDATE_TOL
    # and tests/test_gpu_parity_sizes.py)
    assert abs(got[0] - want[0]) < 1e-3 * want[0], (got, want)
    assert np.allclose(got[:3], want[:3], rtol=1e-2), (got, want)
    assert np.allclose(got, want, rtol=1e-1), (got, want)
    assert [t for t, _, _ in rec.scalars][-1] == "train/total_loss_epoch"
    assert abs(rec.scalars[-1][1] - sum(got)) < 1e-4 * sum(got)
    assert rec.images == 11 and rec.checkpoints == [(["best_pred", "epoch", "optimizer", "state_dict"], False)]
    assert np.allclose([pg["lr"] for pg in opt.param_groups], [zo.poly_lr(1e-3, 11, 0, 12, 2), 10 * zo.poly_lr(1e-3, 11, 0, 12, 2)],
                       rtol=1e-12)
    assert int(net.backbone.bn1.num_batches_tracked) == 11

    # ---- (2) the reference's recorded trajectory (default init, dropout live, lr 1e-5)
    torch.manual_seed(1)
    net = DeepLab(num_classes=21, pretrained=False)
    model = _Passthrough(net.to(dev))
    opt = torch.optim.SGD(_groups(net, 1e-5), momentum=0.9, weight_decay=5e-4, nesterov=False)
    trainer, rec = _make_trainer(model, opt, SegmentationLosses(cuda=True).build_loss("ce"), _supervised_loader(11, 2, 65, 100),
                                 LR_Scheduler("poly", 1e-5, 2, 11, verbose=False), cuda=True)
    trainer.training(0)
    got = np.array([v for tag, v, _ in rec.scalars if tag == "train/total_loss_iter"])
    print("supervised trajectory:", got, g["losses"])
    assert np.allclose([pg["lr"] for pg in opt.param_groups], g["final_lr"], rtol=1e-9)
    assert int(net.backbone.bn1.num_batches_tracked) == int(g["nbt"]) == 11
    # a LOOP check, not an arithmetic one: the reference's recording has live dropout drawn from torch's generator, ours from the
    # counter hash -- different masks, the same distribution; what is pinned is the level of the eleven losses, the schedule and
    # the counters above
    assert np.abs(got - g["losses"]).max() < 0.15 and abs(got.mean() - g["losses"].mean()) < 0.05, (got, g["losses"])


def test_gmmn_trainer_training_epoch(dev):
    """`GMMNTrainer.training(epoch, args)` (train_pascal_GMMN.py:134-311 surface): loader of dict batches, scheduler before
    every step, two scalars per iteration, one batch of lookahead for the pipelined feature pass; equals driving GMMNStep
    by hand on the same seeds."""
    import zs3_oracle as zo
    from zs3_amd import functional as Fz
    from zs3_amd.gmmn_trainer import GMMNStep, GMMNTrainer
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.modeling.gmmn import GMMNnetwork
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    from zs3_amd.utils.lr_scheduler import LR_Scheduler
    seen = [c for c in range(21) if c not in (10, 14)]
    args = types.SimpleNamespace(seen_classes_idx_metric=seen, unseen_classes_idx_metric=[10, 14], noise_dim=300,
                                 embed_dim=300, feature_dim=256, batch_size_generator=128, real_seen_features=True)
    w = torch.ones(21, device=dev)
    w[[10, 14]] = 100.0
    loader = []
    for it in range(2):
        b = zo.make_synthetic_batch(4, 65, seed=200 + it, with_label_emb=True)
        loader.append({k: b[k] for k in ("image", "label", "label_emb")})
    loader.append({k: v[:1] for k, v in loader[0].items()})        # single-sample batch: skipped

    def build():
        torch.manual_seed(1)
        Fz.manual_seed(5)
        net = DeepLab(num_classes=21, pretrained=False).to(dev).train()
        gen = GMMNnetwork(300, 300, 256, 256).to(dev).train()
        opt = SGD(_groups(net, 0.007), momentum=0.9, weight_decay=5e-4)
        return net, gen, opt, Adam(gen.parameters(), lr=2e-4), SegmentationLosses(weight=w, cuda=True).build_loss("ce")

    net, gen, opt, opt_g, crit = build()
    rec = _Recorder()
    tr = GMMNTrainer(args, _Passthrough(net), gen, opt, opt_g, crit, loader, LR_Scheduler("poly", 0.007, 2, 3, verbose=False),
                     writer=rec, noise="cpu")
    torch.manual_seed(31)
    total = tr.training(0, args)
    c_it = [v for t, v, _ in rec.scalars if t == "train/total_loss_iter"]
    g_it = [v for t, v, _ in rec.scalars if t == "train/generator_loss"]
    assert len(c_it) == len(g_it) == 2 and abs(total - sum(c_it)) < 1e-6 * total
    assert [s for t, _, s in rec.scalars if t == "train/total_loss_iter"] == [0, 1]
    net2, gen2, opt2, opt_g2, crit2 = build()
    step = GMMNStep(net2, gen2, opt2, opt_g2, crit2, seen=seen, unseen=[10, 14], noise="cpu")
    sched = LR_Scheduler("poly", 0.007, 2, 3, verbose=False)
    torch.manual_seed(31)
    # the trainer looks one batch ahead: batch 1's feature pass is started next to batch 0's generator loop
    images = [loader[0]["image"].to(dev), loader[1]["image"].to(dev)]
    for i in range(2):
        sched(opt2, i, 0, 0.0)
        gl, cl, _ = step(images[i], loader[i]["label"].to(dev), loader[i]["label_emb"].to(dev),
                         next_image=images[1] if i == 0 else None)
        assert abs(gl - g_it[i]) <= 1e-6 * abs(gl) and abs(cl - c_it[i]) <= 1e-6 * abs(cl)
    for a, b in zip(gen.parameters(), gen2.parameters()):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------- GMMNStep: configs[3], table=
def _gmmn_pair(dev, classes, unseen, pool_bn=True):
    import zs3_oracle as zo
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.modeling.gmmn import GMMNnetwork
    torch.manual_seed(1)
    m = DeepLab(num_classes=classes, pretrained=False, global_avg_pool_bn=pool_bn)
    _tame(m)
    ref = zo.DeepLab(num_classes=classes, pretrained=False, global_avg_pool_bn=pool_bn)
    ref.load_state_dict(m.state_dict())
    torch.manual_seed(2)
    gen = GMMNnetwork(300, 300, 256, 256)
    gen_r = zo.GMMNnetwork(300, 300, 256, 256)
    gen_r.load_state_dict(gen.state_dict())
    _no_dropout(m, ref, gen, gen_r)
    return m.to(dev).train(), ref.train(), gen.to(dev).train(), gen_r.train()


def _run_gmmn_vs_oracle(dev, classes, unseen, use_table, steps=2):
    import zs3_oracle as zo
    from zs3_amd.gmmn_trainer import GMMNStep
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    seen = [c for c in range(classes) if c not in unseen]
    m, ref, gen, gen_r = _gmmn_pair(dev, classes, unseen, pool_bn=classes == 21)
    w = torch.ones(classes)
    w[list(unseen)] = 100.0
    opt, opt_g = SGD(_groups(m, 0.007), momentum=0.9, weight_decay=5e-4), Adam(gen.parameters(), lr=2e-4)
    opt_r = torch.optim.SGD(_groups(ref, 0.007), momentum=0.9, weight_decay=5e-4)
    opt_gr = torch.optim.Adam(gen_r.parameters(), lr=2e-4)
    step = GMMNStep(m, gen, opt, opt_g, SegmentationLosses(weight=w.to(dev), cuda=True).build_loss("ce"), seen=seen,
                    unseen=list(unseen), noise="cpu")
    for it in range(steps):
        b = zo.make_synthetic_batch(4, 65, num_classes=classes, unseen=tuple(unseen), seed=400 + it, with_label_emb=True)
        torch.manual_seed(17 + it)
        gl_r, cl_r = zo.gmmn_step(ref, gen_r, opt_r, opt_gr, zo.SegmentationLosses(weight=w).build_loss("ce"),
                                  zo.GMMNLoss().build_loss(), b["image"], b["label"], b["label_emb"], seen=seen,
                                  unseen=list(unseen))
        torch.manual_seed(17 + it)
        if use_table:   # the [C, 300] table itself; label_emb (5 GB per 513x513 batch in the reference) is never built
            gl, cl, out = step(b["image"].to(dev), b["label"].to(dev), table=b["table"].to(dev))
        else:
            gl, cl, out = step(b["image"].to(dev), b["label"].to(dev), b["label_emb"].to(dev))
        assert abs(gl - gl_r) < 2e-3 * abs(gl_r), (it, gl, gl_r)
        assert abs(cl - cl_r) < 1e-3 * abs(cl_r), (it, cl, cl_r)
        assert out.shape == (4, classes, 65, 65)
    for (k, p), (_, pr) in zip(gen.named_parameters(), gen_r.named_parameters()):
        # Adam moves an element whose gradient is ~0 by +-lr on the sign of rounding noise: the worst element is bounded by
        # a few of the ~50-80 updates taken (3e-2 of the largest weight, measured 2.3e-2 at 60 classes), the mean tightly
        assert rel(p, pr) < 3e-2, k
        assert ((p.detach().cpu() - pr.detach()).abs().mean() / pr.detach().abs().mean()).item() < 2e-3, k
    assert rel(m.decoder.pred_conv.weight, ref.decoder.pred_conv.weight) < 2e-3
    assert rel(m.decoder.pred_conv.bias, ref.decoder.pred_conv.bias) < 2e-3
    return step


def test_gmmn_step_60_classes_vs_oracle(dev):
    """BASELINE configs[3]: train_context_GMMN.py = the GMMN step on Pascal-Context's 60 classes (59 + background,
    datasets/context.py:22; global_avg_pool_bn=False like the context scripts), two unseen classes, against the oracle."""
    step = _run_gmmn_vs_oracle(dev, 60, (5, 17), use_table=False)
    assert step.last_updates > 0


def test_gmmn_step_embedding_table_vs_oracle(dev):
    """`GMMNStep(table=)` looks the class embeddings up on the device at feature resolution; the oracle is
    fed the reference's materialised label_emb tensor (datasets/base.py:45-51).  Same losses, same updates."""
    _run_gmmn_vs_oracle(dev, 21, (10, 14), use_table=True)


def test_more_updates_than_the_staging_ring_had_rows(dev):
    """The pinned ring of sample indices is sized from the step's own update count (a 60-class batch can exceed
    any fixed size before the step's single host sync)."""
    import zs3_oracle as zo
    from zs3_amd.gmmn_trainer import GMMNStep
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.modeling.gmmn import GMMNnetwork
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    torch.manual_seed(1)
    m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
    gen = GMMNnetwork(300, 300, 256, 256).to(dev).train()
    step = GMMNStep(m, gen, SGD(_groups(m, 0.007), momentum=0.9), Adam(gen.parameters(), lr=2e-4),
                    SegmentationLosses(cuda=True).build_loss("ce"), seen=[c for c in range(21) if c not in (10, 14)],
                    unseen=[10, 14], noise="device")
    b = zo.make_synthetic_batch(3, 65, seed=9, with_label_emb=False)
    step(b["image"].to(dev), b["label"].to(dev), table=b["table"].to(dev))
    step._st["ring"] = torch.zeros((2, 128), dtype=torch.int64).pin_memory()      # far fewer rows than updates
    gl, cl, _ = step(b["image"].to(dev), b["label"].to(dev), table=b["table"].to(dev))
    assert step.last_updates > 2 and step._st["ring"].shape[0] >= step.last_updates
    assert np.isfinite(gl) and np.isfinite(cl)


def _u01_host(seed, idx):
    """numpy twin of csrc/common.h::u01 (splitmix64-style counter hash -> 24-bit uniform in [0, 1))"""
    import numpy as np
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx.astype(np.uint64) + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def test_device_noise_path_vs_oracle_with_the_devices_own_draws(dev, monkeypatch):
    """the benched GMMN path (noise="device": hipGraph-captured 7-launch update, sample indices from one
    up-front draw, counter-hash noise keyed on the sampled pixel) was only property-checked, because its random numbers are
    not the reference's CPU stream.  Here the oracle's torch.rand / torch.randint calls (train_pascal_GMMN.py:216,229 order:
    per image, per class) are answered with the device path's OWN draws -- the indices replayed from the same CPU seed, the
    noise rows recomputed on the host from the hash (seed_base + update * 2^24, key = within-class pixel index) -- so the
    two trajectories must agree to kernel accuracy: per-step generator and classifier losses, generator weights after the
    Adam updates, pred_conv after the SGD step.  Seen-only labels (every update is a sampled MMD update); dropout off."""
    import numpy as np
    import zs3_oracle as zo
    from zs3_amd import functional as Fz
    from zs3_amd.gmmn_trainer import GMMNStep
    from zs3_amd.modeling.gmmn import GMMNnetwork
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    seen = [c for c in range(21) if c not in (10, 14)]
    from zs3_amd.modeling.deeplab import DeepLab
    torch.manual_seed(1)
    m = DeepLab(num_classes=21, pretrained=False)
    _tame(m)
    ref = zo.DeepLab(num_classes=21, pretrained=False)
    ref.load_state_dict(m.state_dict())
    _no_dropout(m, ref)
    torch.manual_seed(9)
    gen = GMMNnetwork(300, 300, 256, 256)
    gen_r = zo.GMMNnetwork(300, 300, 256, 256)
    gen_r.load_state_dict(gen.state_dict())
    for mod in list(gen.modules()) + list(gen_r.modules()):
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    m, gen = m.to(dev).train(), gen.to(dev).train()
    ref.train()
    gen_r.train()
    w = torch.ones(21)
    w[[10, 14]] = 100.0

    def groups(mod, lr):
        return [{"params": mod.get_1x_lr_params(), "lr": lr}, {"params": mod.get_10x_lr_params(), "lr": lr * 10}]

    opt, opt_g = SGD(groups(m, 0.007), momentum=0.9, weight_decay=5e-4), Adam(gen.parameters(), lr=2e-4)
    opt_r, opt_gr = torch.optim.SGD(groups(ref, 0.007), momentum=0.9, weight_decay=5e-4), torch.optim.Adam(gen_r.parameters(), lr=2e-4)
    Fz.manual_seed(123)
    step = GMMNStep(m, gen, opt, opt_g, SegmentationLosses(weight=w.to(dev), cuda=True).build_loss("ce"), seen=seen,
                    unseen=[10, 14], noise="device")
    bsg, nz = 128, 300
    updates_done = 0
    for it in range(2):
        b = zo.make_synthetic_batch(3, 65, seed=400 + it, with_label_emb=True)     # images 0..2: no unseen class
        torch.manual_seed(31 + it)
        gl, cl, _ = step(b["image"].to(dev), b["label"].to(dev), table=b["table"].to(dev))
        assert step._st["table_mode"]
        seed_base = int(step._st["seed_base"])
        # ---- the device path's draws, replayed: (image, class) pairs in the reference's loop order
        fh = fw = step._st["shape"][1] ** 0.5
        fh = fw = int(round(fh))
        tgt = torch.nn.functional.interpolate(b["label"][:, None], size=(fh, fw), mode="nearest").reshape(3, -1)
        pairs = [(i, float(c), int((tgt[i] == c).sum())) for i in range(3) for c in torch.unique(tgt[i]) if float(c) != 255]
        torch.manual_seed(31 + it)
        n_c = torch.tensor([p[2] for p in pairs], dtype=torch.float64)[:, None]
        idx_all = torch.minimum((torch.rand((len(pairs), bsg), dtype=torch.float64) * n_c).long(), n_c.long() - 1)
        answers = []
        for k, (i, c, n) in enumerate(pairs):
            seed_k = (seed_base + (updates_done + k) * (1 << 24)) % (1 << 64)
            keys = np.arange(n, dtype=np.uint64)[:, None] * np.uint64(nz) + np.arange(nz, dtype=np.uint64)[None, :]
            answers.append(("rand", torch.from_numpy(_u01_host(seed_k, keys))))
            answers.append(("randint", idx_all[k].clone()))
        updates_done += len(pairs)
        it_ans = iter(answers)

        def fake_rand(*size, **kw):
            kind, val = next(it_ans)
            assert kind == "rand" and tuple(val.shape) == tuple(size[0] if isinstance(size[0], (tuple, list)) else size)
            return val

        def fake_randint(*a, **kw):
            kind, val = next(it_ans)
            assert kind == "randint" and kw.get("high", None) is not None and int(val.max()) < kw["high"]
            return val

        monkeypatch.setattr(torch, "rand", fake_rand)
        monkeypatch.setattr(torch, "randint", fake_randint)
        try:
            gl_r, cl_r = zo.gmmn_step(ref, gen_r, opt_r, opt_gr, zo.SegmentationLosses(weight=w).build_loss("ce"),
                                      zo.GMMNLoss().build_loss(), b["image"], b["label"], b["label_emb"], seen=seen, unseen=[10, 14])
        finally:
            monkeypatch.undo()
        assert next(it_ans, None) is None, "the oracle consumed fewer draws than the device path made"
        assert abs(gl - gl_r) < 2e-3 * abs(gl_r), (it, gl, gl_r)
        assert abs(cl - cl_r) < 1e-3 * abs(cl_r), (it, cl, cl_r)
    assert int(step.last_updates) == len(pairs)
    for (k, p), (_, pr) in zip(gen.named_parameters(), gen_r.named_parameters()):
        e_mean = ((p.detach().cpu() - pr.detach()).abs().mean() / pr.detach().abs().mean()).item()
        assert rel(p, pr) < 2e-2 and e_mean < 1.5e-3, (k, rel(p, pr), e_mean)
    assert rel(m.decoder.pred_conv.weight, ref.decoder.pred_conv.weight) < 2e-3
