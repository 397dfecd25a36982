"""Single-GPU exercise of the multi-GPU plumbing on real RCCL: a one-rank NCCL process group runs the bucketed gradient
all-reduce from the wgrad side stream, the globally normalised CE and the SyncBN statistics path.  With one rank every
collective is the identity, so the step must reproduce the plain single-process step (up to fp64 re-association in the SyncBN sums)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def one_rank_group(dev):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield True
    import zs3_amd.parallel as par
    torch.cuda.synchronize()
    par.native_shutdown()
    dist.destroy_process_group()


def _step(dev, ddp):
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.optim import SGD
    from zs3_amd.parallel import GradSync, broadcast_parameters, enable_sync_bn
    from zs3_amd.utils.loss import SegmentationLosses
    from zs3_amd.utils.synthetic import make_batch
    torch.manual_seed(1)
    m = DeepLab(num_classes=21, pretrained=False, sync_bn=True)
    for name, mod in m.named_modules():
        if name.endswith("bn3"):
            mod.weight.data.fill_(0.1)
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m = m.to(dev).train()
    sync = None
    import zs3_amd.parallel as par
    par.FORCE_COLLECTIVES = ddp
    if ddp:
        broadcast_parameters(m)
        assert enable_sync_bn(m) == 113
        sync = GradSync(list(m.parameters()), bucket_mb=16.0, force=True)
        assert len(sync.buckets) > 4
    groups = [{"params": m.get_1x_lr_params(), "lr": 1e-3}, {"params": m.get_10x_lr_params(), "lr": 1e-2}]
    opt = SGD(groups, momentum=0.9, weight_decay=5e-4)
    crit = SegmentationLosses(cuda=True, group=True if ddp else None).build_loss("ce")
    b = make_batch(4, 97, seed=3, device=dev)
    losses = []
    for _ in range(2):
        opt.zero_grad()
        loss = crit(m(b["image"]), b["label"])
        loss.backward()
        opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    if sync is not None:
        assert sync.bytes_reduced == 2 * 4 * sum(p.numel() for p in m.parameters())
        # zero-copy buckets: the conv weights' .grad ARE their bucket slices (written there by the wgrad kernels)
        convs = [p for p in m.parameters() if p.dim() == 4 and p.grad is not None and p is not m.backbone.conv1.weight]
        assert sum(sync._in_place(p) for p in convs) == len(convs) > 100
        sync.remove()
    par.FORCE_COLLECTIVES = False
    return losses, {k: v.detach().clone() for k, v in m.state_dict().items()}


def test_one_rank_rccl_step_equals_plain_step(dev, one_rank_group):
    import zs3_amd.parallel as par
    l0, s0 = _step(dev, ddp=False)
    assert par.native_available() and not par._native_comms
    l1, s1 = _step(dev, ddp=True)
    # round 6: the collectives were issued by libzs3hip.so itself (csrc/comm.hip), one communicator per issuing stream -- the main
    # stream (SyncBN, CE, range flag), the two ASPP lanes (their SyncBN layers) and the weight-gradient stream (the buckets)
    assert len(par._native_comms) == 4, par._native_comms
    par.NATIVE_RCCL = False          # the same step with every collective through torch.distributed (the round-5 path)
    try:
        l2, s2 = _step(dev, ddp=True)
    finally:
        par.NATIVE_RCCL = True
    assert max(abs(a - b) for a, b in zip(l1, l2)) < 1e-6 * abs(l1[0]), (l1, l2)
    for k in s1:
        if s1[k].dtype.is_floating_point:
            assert (s1[k] - s2[k]).abs().max().item() <= 1e-6 * max(1.0, s1[k].abs().max().item()), k
    # one rank: every collective is the identity; the SyncBN path only re-associates fp64 sums
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-5, (l0, l1)
    for k in s0:
        if s0[k].dtype.is_floating_point:
            err = (s0[k] - s1[k]).abs().max().item()
            # (the plain run sums the BN-backward statistics in the dgrad epilogues, the SyncBN run in a separate pass)
            assert err <= 1e-5 + 1e-4 * s0[k].abs().max().item(), (k, err)


def test_native_probe_and_its_fallback(dev, one_rank_group, monkeypatch):
    """The first multi-rank use of the library's own communicator checks itself (parallel.native_probe: rank r contributes r + 1,
    every rank must read the triangular number; verdicts MIN-reduced through torch.distributed).  One rank exercises the code:
    the good path answers True; a communicator that cannot be created, a wrong sum and a collective that never completes all
    answer False with a warning, after which the collectives of the step go through torch.distributed."""
    import zs3_amd.parallel as par
    assert par.native_probe() is True

    # a wrong sum: the probe believes the group has three ranks (the communicator of the first probe is cached: nothing is created)
    with monkeypatch.context() as mp:
        mp.setattr(dist, "get_world_size", lambda g=None: 3)
        with pytest.warns(UserWarning, match="wrong sum"):
            assert par.native_probe() is False
    def broken(*a, **k):
        raise RuntimeError("zs3_comm_create failed (injected)")
    with monkeypatch.context() as mp:
        mp.setattr(par, "native_comm", broken)
        with pytest.warns(UserWarning, match="torch.distributed instead"):
            assert par.native_probe() is False
    assert not par._native_comms                           # (a failed probe destroys the group's communicators)
    # a collective that never completes: the event never reports done, the communicator is aborted
    with monkeypatch.context() as mp:
        mp.setattr(par, "PROBE_TIMEOUT_S", 0.05)
        mp.setattr(torch.cuda.Event, "query", lambda self: False)
        with pytest.warns(UserWarning, match="did not complete"):
            assert par.native_probe() is False
    assert not par._native_comms
    # the verdict is what native_available caches for a multi-rank group
    with monkeypatch.context() as mp:
        par._native_backend.clear()
        mp.setattr(par, "native_probe", lambda g=None: False)
        real = dist.get_world_size
        mp.setattr(dist, "get_world_size", lambda g=None: 2)
        assert par.native_available() is False
        mp.setattr(dist, "get_world_size", real)
        t = torch.ones(8, device=dev)
        assert par.native_allreduce(t) is False            # callers take the torch.distributed branch
    par._native_backend.clear()
    assert par.native_available() is True


def test_one_rank_rccl_step_replays_from_a_plan(dev, one_rank_group):
    """The N > 1 code path as a recorded plan: with the library's own collectives (zs3_allreduce, zs3_bn_sync_exchange) the SyncBN /
    GradSync / global-CE step records and replays like the plain one -- bit-identical to its eager self, collectives included in
    the recorded ops."""
    import zs3_amd.parallel as par
    from zs3_amd import functional as Fz
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.optim import SGD
    from zs3_amd.plan import StepPlan
    from zs3_amd.utils.loss import SegmentationLosses
    from zs3_amd.utils.synthetic import make_batch
    par.FORCE_COLLECTIVES = True
    try:
        outs = []
        for use_plan in (False, True):
            torch.manual_seed(1)
            m = DeepLab(num_classes=21, pretrained=False, sync_bn=True)
            for name, mod in m.named_modules():
                if name.endswith("bn3"):
                    mod.weight.data.fill_(0.1)
            m = m.to(dev).train()
            groups = [{"params": m.get_1x_lr_params(), "lr": 1e-3}, {"params": m.get_10x_lr_params(), "lr": 1e-2}]
            opt = SGD(groups, momentum=0.9, weight_decay=5e-4)
            crit = SegmentationLosses(cuda=True).build_loss("ce")       # group="auto": global under FORCE_COLLECTIVES
            Fz.manual_seed(5)
            step = StepPlan(m, crit, opt, enabled=use_plan)
            losses = []
            for i in range(6):
                b = make_batch(2, 97, seed=70 + i, device=dev)
                losses.append(step(b["image"], b["label"])[1].detach().clone())
            torch.cuda.synchronize()
            sync = getattr(m, "_zs3_grad_sync", None)
            assert sync is not None                                      # armed by the model's first training forward
            assert all(sync._in_place(p) for p in sync.params if p.grad is not None)    # every gradient lives in its bucket: no pack copies
            names = step._plan.names() if step._plan is not None else []
            outs.append((torch.stack(losses).cpu(), {k: v.detach().clone() for k, v in m.state_dict().items()},
                         (step.eager_calls, step.recordings, step.replays), names))
            par.disarm_data_parallel(m)
            step.close()
        (le, se, ce, _), (lp, sp, cp, names) = outs
        assert ce == (6, 0, 0) and cp == (2, 1, 3), (ce, cp)
        assert torch.equal(le, lp), (le, lp)
        assert not [k for k in se if not torch.equal(se[k], sp[k])]
        # one exchange per BatchNorm and pass (the backward one also writes this rank's dgamma / dbeta: no per-rank finalize launch)
        assert names.count("zs3_bn_sync_exchange") == 113 and names.count("zs3_bn_sync_exchange_bwd") == 113
        assert names.count("zs3_bn_bwd_finalize") == 113
        assert names.count("zs3_allreduce") >= 3     # buckets + CE + range flag
    finally:
        par.FORCE_COLLECTIVES = False


def _gmmn_steps(dev, ddp):
    from zs3_amd.gmmn_trainer import GMMNStep
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.modeling.gmmn import GMMNnetwork
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    from zs3_amd.utils.synthetic import make_batch
    import zs3_amd.parallel as par
    from zs3_amd import functional as Fz
    seen = [c for c in range(21) if c not in (10, 14)]
    torch.manual_seed(4)
    Fz.manual_seed(77)          # device dropout masks (decoder, generator) are a function of this counter
    m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
    gen = GMMNnetwork(300, 300, 256, 256).to(dev).train()
    w = torch.ones(21, device=dev)
    w[[10, 14]] = 100.0
    groups = [{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}]
    opt, opt_g = SGD(groups, momentum=0.9, weight_decay=5e-4), Adam(gen.parameters(), lr=2e-4)
    par.FORCE_COLLECTIVES = ddp
    crit = SegmentationLosses(weight=w, cuda=True, group=True if ddp else None).build_loss("ce")
    step = GMMNStep(m, gen, opt, opt_g, crit, seen=seen, unseen=[10, 14], noise="cpu", group=True if ddp else None)
    assert step.grad_reduce == ("sum" if ddp else "mean")
    table = torch.nn.functional.normalize(torch.randn(21, 300, generator=torch.Generator().manual_seed(5)), dim=1).to(dev)
    out = []
    for it in range(2):
        b = make_batch(4, 65, 21, [10, 14], seed=300 + it, device=dev)
        torch.manual_seed(21 + it)
        gl, cl, _ = step(b["image"], b["label"], table=table)
        out.append((gl, cl))
    torch.cuda.synchronize()
    par.FORCE_COLLECTIVES = False
    if ddp:   # two exchanges per step: 219,648 generator parameters + pred_conv (21*256 + 21) gradients
        assert step.bytes_reduced == 2 * 4 * (219648 + 21 * 256 + 21)
    return out, [p.detach().clone() for p in gen.parameters()], m.decoder.pred_conv.weight.detach().clone()


def test_one_rank_rccl_gmmn_step_equals_plain_step(dev, one_rank_group):
    """GMMNStep(group=...): generator-parameter averaging + pred_conv gradient reduction through RCCL; with one rank both
    are the identity, so losses and every updated weight must equal the single-process step (to fp32 rounding of the
    globally normalised CE)."""
    l0, g0, p0 = _gmmn_steps(dev, ddp=False)
    l1, g1, p1 = _gmmn_steps(dev, ddp=True)
    for (ga, ca), (gb, cb) in zip(l0, l1):
        assert abs(ga - gb) <= 1e-6 * abs(ga) and abs(ca - cb) <= 1e-6 * abs(ca), (l0, l1)
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)
    assert torch.allclose(p0, p1, rtol=1e-6, atol=1e-8)
