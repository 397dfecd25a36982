"""The recorded launch plan (zs3_amd/plan.py, csrc/plan.hip; VERDICT r5 "next" #2): a training step replayed from C is
bit-identical to the eager step -- same losses, same parameters after N steps -- with live dropout, a learning-rate schedule that
changes every iteration and a fresh input batch per step, in both storage modes; whatever changes the step drops the plan."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _setup(dev, classes=21, seed=1):
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.optim import SGD
    from zs3_amd.utils.loss import SegmentationLosses
    torch.manual_seed(seed)
    model = DeepLab(num_classes=classes, pretrained=False, sync_bn=False)
    for name, mod in model.named_modules():
        if name.endswith("bn3"):
            mod.weight.data.fill_(0.1)
    model = model.to(dev).train()
    groups = [{"params": model.get_1x_lr_params(), "lr": 0.007}, {"params": model.get_10x_lr_params(), "lr": 0.07}]
    opt = SGD(groups, momentum=0.9, weight_decay=5e-4, nesterov=False)
    w = torch.ones(classes, device=dev)
    w[[10, 14]] = 100.0
    crit = SegmentationLosses(weight=w, cuda=True).build_loss("ce")
    return model, opt, crit


def _batches(dev, n, size, steps, classes=21):
    from zs3_amd.utils.synthetic import make_batch
    return [make_batch(n, size, classes, [10, 14], seed=50 + i, device=dev) for i in range(steps)]


def _run(dev, steps, size, use_plan, storage=torch.float32, n=2):
    from zs3_amd import functional as Fz
    from zs3_amd import ops
    from zs3_amd.plan import StepPlan
    from zs3_amd.utils.lr_scheduler import LR_Scheduler
    ops.set_storage(storage)
    try:
        model, opt, crit = _setup(dev)
        Fz.manual_seed(1234)
        sched = LR_Scheduler("poly", 0.007, 1, steps, verbose=False)
        step = StepPlan(model, crit, opt, enabled=use_plan)
        losses = []
        for i, b in enumerate(_batches(dev, n, size, steps)):
            sched(opt, i, 0, 0.0)
            pred, loss = step(b["image"], b["label"])
            losses.append(loss.detach().clone())
            assert pred.shape == (n, 21, size, size)
        torch.cuda.synchronize()
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        mom = [opt.state[p]["momentum_buffer"].detach().clone() for g in opt.param_groups for p in g["params"]]
        counts = (step.eager_calls, step.recordings, step.replays)
        names = step._plan.names() if step._plan is not None else []
        step.close()
        return torch.stack(losses).cpu(), state, mom, counts, names
    finally:
        ops.set_storage(torch.float32)


@pytest.mark.parametrize("storage", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_replayed_steps_equal_eager_steps_bit_for_bit(dev, storage):
    """9 iterations: eager everywhere against 2 eager + 1 recorded + 6 replayed; live dropout (three seeds per step, drawn in the
    eager order), poly schedule (a new lr for both groups before every step), a different batch in different storage per step."""
    steps = 9
    le, se, me, ce, _ = _run(dev, steps, 97, use_plan=False, storage=storage)
    lp, sp, mp, cp, names = _run(dev, steps, 97, use_plan=True, storage=storage)
    assert ce == (steps, 0, 0) and cp == (2, 1, steps - 3), (ce, cp)
    print(f"\n[plan {storage}] losses eager {le.tolist()}\n[plan {storage}] losses plan  {lp.tolist()}")
    assert torch.equal(le, lp)
    bad = [k for k in se if not torch.equal(se[k], sp[k])]
    assert not bad, bad[:8]
    assert all(torch.equal(a, b) for a, b in zip(me, mp))
    # what a step is made of: every launch of the library, the cross-stream waits, one optimizer launch, one plane refresh
    assert len(names) > 600 and names.count("zs3_sgd_multi_g") == 1 and names.count("zs3_prep_weight_multi") == 1
    assert names.count("zs3_stream_wait") >= 100 and names.count("zs3_ce_fwd") == 1 and names.count("zs3_ce_bwd") == 1
    assert names[-1] == "zs3_prep_weight_multi"
    print(f"[plan {storage}] {len(names)} recorded ops: " + ", ".join(f"{n} x{names.count(n)}" for n in sorted(set(names))))


@pytest.mark.parametrize("storage", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_plan_verifies_against_poisoned_memory(dev, storage):
    """StepPlan.verify: the pool's free memory (the plan's activations and workspaces between replays) is filled with NaN patterns,
    the plan is replayed and the same step is run eagerly from the same state and seeds -- loss, prediction, all gradients and all
    persistent tensors must agree bit for bit.  A launch the plan misses (a fill / copy made by the tensor library while recording)
    cannot hide behind last step's bytes."""
    from zs3_amd import functional as Fz
    from zs3_amd import ops
    from zs3_amd.plan import StepPlan
    ops.set_storage(storage)
    try:
        model, opt, crit = _setup(dev)
        Fz.manual_seed(7)
        step = StepPlan(model, crit, opt)
        bs = _batches(dev, 2, 129, 5)
        for b in bs[:4]:
            step(b["image"], b["label"])
        assert step.recordings == 1 and step.replays == 1
        bad = step.verify(bs[4]["image"], bs[4]["label"], poison=True)
        assert not bad, bad[:10]
        assert torch.isfinite(torch.stack([p.detach().abs().max() for p in model.parameters()])).all()
    finally:
        ops.set_storage(torch.float32)


def test_what_changes_the_step_drops_the_plan(dev):
    from zs3_amd import functional as Fz
    from zs3_amd import ops
    from zs3_amd.plan import StepPlan
    model, opt, crit = _setup(dev)
    Fz.manual_seed(3)
    step = StepPlan(model, crit, opt)
    b65, b97 = _batches(dev, 2, 65, 1)[0], _batches(dev, 2, 97, 1)[0]

    def run(b, k):
        for _ in range(k):
            out = step(b["image"], b["label"])
        return out

    run(b65, 4)
    assert (step.eager_calls, step.recordings, step.replays) == (2, 1, 1)
    run(b97, 1)                                  # another input shape: eager again, then a new recording
    assert step.eager_calls == 3 and step._plan is None
    run(b97, 3)
    assert (step.recordings, step.replays) == (2, 2)
    model.backbone.layer4[0].bn1.eval()          # a train / eval flip anywhere in the model
    run(b97, 1)
    assert step._plan is None
    model.backbone.layer4[0].bn1.train()
    run(b97, 3)
    assert step.recordings == 3 and step._plan is not None
    opt.param_groups[0]["momentum"] = 0.8        # a hyper-parameter that is baked into the recorded launch
    run(b97, 1)
    assert step._plan is None
    opt.param_groups[0]["momentum"] = 0.9
    run(b97, 3)
    assert step.recordings == 4
    ops.set_storage(torch.bfloat16)              # a mode switch (PLAN_EPOCH)
    try:
        run(b97, 1)
        assert step._plan is None
    finally:
        ops.set_storage(torch.float32)
    run(b97, 3)
    assert step.recordings == 5 and step._plan is not None
    with torch.no_grad():                        # no gradient mode, a foreign optimizer: plain eager calls, the plan stays
        pass
    step2 = StepPlan(model, crit, torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9))
    for _ in range(4):
        step2(b65["image"], b65["label"])
    assert (step2.recordings, step2.replays, step2.eager_calls) == (0, 0, 4)
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_plan_entry_points_record_patch_and_replay(dev):
    """the C ABI of the plan on two launches: zs3_affine_act (a dropout seed among its arguments) and zs3_sgd_multi_g (a host array
    among its arguments): record, replay into fresh outputs, patch the seed / the learning rate, replay again"""
    import ctypes
    from zs3_amd import ops
    from zs3_amd._lib import lib
    from zs3_amd.optim import SGD
    from zs3_amd.plan import LaunchPlan
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(64, 32, device=dev, generator=g)
    sc, sh = torch.rand(32, device=dev, generator=g) + 0.5, torch.randn(32, device=dev, generator=g)
    out = torch.empty_like(x)
    plan = LaunchPlan()
    plan.begin()
    ops.affine_act(x, sc, sh, out=out, act=1, drop=(0.5, 1111))
    assert plan.end() == 1 and plan.names() == ["zs3_affine_act"]
    want1 = out.clone()
    want2 = ops.affine_act(x, sc, sh, act=1, drop=(0.5, 2222))
    assert not torch.equal(want1, want2)
    out.zero_()
    plan.replay()
    assert torch.equal(out, want1)
    assert plan.replace_u64(1111, 2222) == 1 and plan.replace_u64(1111, 3333) == 0
    plan.replay()
    assert torch.equal(out, want2)
    x2 = x * 2
    assert plan.replace_ptr(x.data_ptr(), x2.data_ptr()) == 1
    plan.replay()
    assert torch.equal(out, ops.affine_act(x2, sc, sh, act=1, drop=(0.5, 2222)))
    sid = plan.time_ops([0])                     # HIP events around op 0 inside the next replay, on the op's own stream
    plan.replay()
    torch.cuda.synchronize()
    ms = plan.timed_ms(sid, 1)
    assert len(ms) == 1 and 0.0 < ms[0] < 5.0
    plan.replay()                                # one-shot: this replay records nothing new, the set stays readable
    torch.cuda.synchronize()
    assert plan.timed_ms(sid, 1) == ms and lib().zs3_plan_time_ops(plan.handle, (ctypes.c_int * 2)(0, 0), 2) == -3
    assert lib().zs3_plan_patch(plan.handle, 0, 0, (ctypes.c_float * 1)(1.0), 4) == -4       # pointers are not patched as scalars
    assert lib().zs3_plan_patch(plan.handle, 5, 0, (ctypes.c_float * 1)(1.0), 4) == -1
    plan.close()
    # SGD: lr as a launch argument
    p = torch.nn.Parameter(torch.randn(1000, device=dev, generator=g))
    opt = SGD([p], lr=0.1, momentum=0.0)
    p.grad = torch.ones_like(p)
    p0 = p.detach().clone()
    plan = LaunchPlan()
    plan.begin()
    opt.step()
    plan.end()
    assert plan.names() == ["zs3_sgd_multi_g"] and torch.allclose(p.detach(), p0 - 0.1)
    plan.patch(plan.find("zs3_sgd_multi_g"), 6, (ctypes.c_float * 2)(0.25, 0.0))
    plan.replay()
    torch.cuda.synchronize()
    assert torch.allclose(p.detach(), p0 - 0.35)
    plan.close()


def test_aspp_lanes_change_nothing_but_the_schedule(dev):
    """functional lanes (round 6): ASPP's two heavy atrous branches run on side streams entered INSIDE the fused layer (forward and
    backward, waits recorded through zs3_stream_wait) -- the result is bit-identical to the sequential module, logits and every
    gradient, and autograd still sees one stream."""
    from zs3_amd import functional as Fz
    from zs3_amd.utils.loss import SegmentationLosses
    outs = []
    b = _batches(dev, 2, 129, 1)[0]
    for concurrent in (True, False):
        Fz.ASPP_CONCURRENT = concurrent
        try:
            model, opt, crit = _setup(dev)
            Fz.manual_seed(11)
            out = model(b["image"])
            loss = crit(out, b["label"])
            loss.backward()
            torch.cuda.synchronize()
            outs.append((out.detach().clone(), loss.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]))
        finally:
            Fz.ASPP_CONCURRENT = True
    assert not Fz._lanes_open and not Fz._lanes_armed[0]                   # every lane was joined
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert all(torch.equal(a, c) for a, c in zip(outs[0][2], outs[1][2]))


def test_feature_pass_replays_from_a_plan(dev):
    """ForwardPlan (zs3_amd/plan.py): the GMMN step's frozen-backbone feature pass -- train-mode BatchNorm, live dropout, no
    gradients -- recorded on its third call and replayed: features, BatchNorm running statistics and the dropout stream are those
    of the eager calls, bit for bit, on a fresh input tensor every call."""
    from zs3_amd import functional as Fz
    from zs3_amd import ops
    from zs3_amd.plan import ForwardPlan
    res = []
    bs = _batches(dev, 2, 129, 6)
    for use_plan in (False, True):
        model, _, _ = _setup(dev)
        Fz.manual_seed(21)
        fp = ForwardPlan(lambda im: ops.nhwc(model.forward_before_class_prediction(im)), [model], enabled=use_plan)
        feats = []
        with torch.no_grad():
            for b in bs:
                feats.append(fp(b["image"]).clone())
        torch.cuda.synchronize()
        res.append((feats, {k: v.detach().clone() for k, v in model.state_dict().items()}, (fp.eager_calls, fp.recordings, fp.replays)))
        fp.close()
    assert res[0][2] == (6, 0, 0) and res[1][2] == (2, 1, 3), (res[0][2], res[1][2])
    assert all(torch.equal(a, c) for a, c in zip(res[0][0], res[1][0]))
    assert not [k for k in res[0][1] if not torch.equal(res[0][1][k], res[1][1][k])]
    assert not torch.equal(res[1][0][4], res[1][0][5])                      # (live dropout, another batch: the replays do differ)


def test_plan_with_the_reference_scripts_cpu_class_weights_and_with_labels_it_cannot_rebind(dev):
    """(1) The reference's scripts hand the criterion a CPU weight tensor and move it at every call (loss.py:36-37): the device copy
    is made once per tensor version, so the recorded CE launch reads a buffer that stays -- replays equal eager steps.  (2) uint8
    labels are cast by the tensor library in front of the first launch: the step does not read the caller's tensor, the plan
    cannot rebind it -- that configuration stays eager (one attempted recording, then none), results unchanged."""
    from zs3_amd import functional as Fz
    from zs3_amd.plan import StepPlan
    from zs3_amd.utils.loss import SegmentationLosses
    w = torch.ones(21)
    w[[10, 14]] = 100.0
    bs = _batches(dev, 2, 65, 6)
    outs = []
    for use_plan in (False, True):
        model, opt, _ = _setup(dev)
        crit = SegmentationLosses(weight=w, cuda=True).build_loss("ce")            # CPU weights, like train_pascal.py:64-77
        Fz.manual_seed(5)
        step = StepPlan(model, crit, opt, enabled=use_plan)
        losses = [step(b["image"], b["label"])[1].detach().clone() for b in bs]
        torch.cuda.synchronize()
        outs.append((torch.stack(losses).cpu(), (step.eager_calls, step.recordings, step.replays)))
        step.close()
    assert outs[1][1] == (2, 1, 3) and torch.equal(outs[0][0], outs[1][0]), outs
    model, opt, crit = _setup(dev)
    Fz.manual_seed(5)
    step = StepPlan(model, crit, opt)
    for b in bs:
        _, loss = step(b["image"], b["label"].to(torch.uint8))
    torch.cuda.synchronize()
    # 2 settling calls, 1 attempted recording (it ran the step eagerly and found nothing to rebind), 3 plain eager calls
    assert (step.recordings, step.replays, step.eager_calls) == (0, 0, 5) and step._plan is None and torch.isfinite(loss)
