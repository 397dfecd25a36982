"""GPU parity of the latency-shaped GMMN update kernels (csrc/gmmn.hip) against fp64 torch on the host and against the
general kernels they replace (same counter RNG: noise and dropout masks must be bit-identical)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("rows,p_drop", [(128, 0.5), (100, 0.0), (37, 0.3)])
def test_mlp_forward_backward_kernels(dev, rows, p_drop):
    """gmmn.py:17-22 on S sampled rows: Linear(600, 256) + LeakyReLU(0.2) + Dropout + Linear(256, 256) and its backward."""
    from zs3_amd import ops
    from zs3_amd._lib import F, I, P, check, lib, stream
    g = torch.Generator().manual_seed(rows)
    npix, e, nz, hid, d = 500, 300, 300, 256, 256
    emb = torch.randn(npix, e, generator=g).to(dev)
    real = torch.randn(3 * npix, d, generator=g).to(dev)
    w1 = (torch.randn(hid, e + nz, generator=g) / 24.0).to(dev)
    b1 = (torch.randn(hid, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(d, hid, generator=g) / 16.0).to(dev)
    b2 = (torch.randn(d, generator=g) * 0.1).to(dev)
    key = torch.randint(0, 60, (rows,), generator=g).to(dev)            # duplicates on purpose
    pix = torch.randint(0, npix, (rows,), generator=g).to(dev)
    gidx = (pix + npix).contiguous()
    seed_dev = torch.tensor([12345], dtype=torch.int64, device=dev)
    seed_n, seed_d, leak = 987654321, 55555, 0.2
    wp1, wp2 = ops.prep_weight(w1), ops.prep_weight(w2)
    width = e + nz
    x = torch.empty(rows, width, device=dev)
    h = torch.empty(rows, hid, device=dev)
    hd = torch.empty(rows, hid, device=dev)
    check(lib().zs3_gmmn_mlp_fwd1(P(emb), I(e), P(pix), P(key), I(e), I(nz), P(wp1.f_pk), I(wp1.cin_pad // 32), P(b1), P(x),
                                  I(width), P(h), P(hd), I(hid), I(rows), I(hid), F(leak), F(p_drop), ctypes.c_ulonglong(seed_n),
                                  ctypes.c_ulonglong(seed_d), P(seed_dev), stream()), "fwd1")
    x_ref = ops.gather_cat_noise(emb, pix, e, nz, width, rows, seed_n, seed_dev=seed_dev, noise_key=key)
    assert torch.equal(x, x_ref)                                                    # same rows, same noise bits
    h_ref = torch.nn.functional.leaky_relu(x.double().cpu() @ w1.double().cpu().t() + b1.double().cpu(), leak)
    assert rel(h, h_ref) < 2e-5
    hd_ref = ops.dropout(h, p_drop, seed_d, row_idx=key, seed_dev=seed_dev) if p_drop > 0 else h
    assert torch.equal(hd, hd_ref)                                                  # same mask as zs3_dropout
    gen = torch.empty(rows, d, device=dev)
    real_s = torch.empty(rows, d, device=dev)
    check(lib().zs3_gmmn_mlp_fwd2(P(hd), I(hid), P(wp2.f_pk), I(wp2.cin_pad // 32), P(b2), P(gen), I(d), I(rows), I(d), I(hid),
                                  P(real), I(d), P(gidx), P(real_s), stream()), "fwd2")
    assert rel(gen, hd.double().cpu() @ w2.double().cpu().t() + b2.double().cpu()) < 2e-5
    assert torch.equal(real_s, real[gidx])
    dgen = torch.randn(rows, d, generator=g).to(dev)
    dpre = torch.empty(rows, hid, device=dev)
    check(lib().zs3_gmmn_mlp_dgrad(P(dgen), I(d), P(wp2.t_pk), I(wp2.cout_pad // 32), P(h), I(hid), P(key), P(dpre), I(hid),
                                   I(rows), I(hid), I(d), F(leak), F(p_drop), ctypes.c_ulonglong(seed_d), P(seed_dev), stream()),
          "dgrad")
    dhd_ref = (dgen.double().cpu() @ w2.double().cpu()).float().to(dev)
    dpre_ref = ops.dropout_act_bwd(dhd_ref, h, p_drop, seed_d, leak, row_idx=key, seed_dev=seed_dev)
    assert rel(dpre, dpre_ref) < 2e-5
    dw1, db1 = torch.empty(hid, width, device=dev), torch.empty(hid, device=dev)
    dw2, db2 = torch.empty(d, hid, device=dev), torch.empty(d, device=dev)
    check(lib().zs3_gmmn_mlp_wgrad(P(dgen), I(d), P(hd), I(hid), I(d), I(hid), P(dw2), P(db2), P(dpre), I(hid), P(x), I(width),
                                   I(hid), I(width), P(dw1), P(db1), I(rows), stream()), "wgrad")
    assert rel(dw2, dgen.double().cpu().t() @ hd.double().cpu()) < 2e-5
    assert rel(db2, dgen.double().cpu().sum(0)) < 1e-5
    assert rel(dw1, dpre.double().cpu().t() @ x.double().cpu()) < 2e-5
    assert rel(db1, dpre.double().cpu().sum(0)) < 1e-5


def test_fused_update_equals_general_kernels(dev):
    """GMMNStep(fused_mlp=True) against the same step on the general conv / elementwise kernels: same RNG stream, same
    arithmetic class (bf16x3, fp32 accumulate), different summation order -> losses to 2e-3, weights to Adam's noise floor."""
    import zs3_oracle as zo
    from zs3_amd import functional as Fz
    from zs3_amd.gmmn_trainer import GMMNStep
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.modeling.gmmn import GMMNnetwork
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    seen = [c for c in range(21) if c not in (10, 14)]
    b = zo.make_synthetic_batch(4, 65, seed=77, with_label_emb=False)
    out = []
    for fused in (True, False):
        for noise in ("device", "cpu"):
            torch.manual_seed(1)
            Fz.manual_seed(9)
            m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
            gen = GMMNnetwork(300, 300, 256, 256).to(dev).train()
            groups = [{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}]
            step = GMMNStep(m, gen, SGD(groups, momentum=0.9, weight_decay=5e-4), Adam(gen.parameters(), lr=2e-4),
                            SegmentationLosses(cuda=True).build_loss("ce"), seen=seen, unseen=[10, 14], noise=noise,
                            fused_mlp=fused)
            torch.manual_seed(3)
            gl, cl, _ = step(b["image"].to(dev), b["label"].to(dev), table=b["table"].to(dev))
            out.append((fused, noise, gl, cl, [p.detach().clone() for p in gen.parameters()], step.last_updates))
    for noise in ("device", "cpu"):
        a = next(o for o in out if o[0] and o[1] == noise)
        c = next(o for o in out if not o[0] and o[1] == noise)
        assert a[5] == c[5] > 20
        if noise == "cpu":   # same sample indices, same noise: ~50 dependent Adam updates amplify the summation order only
            assert abs(a[2] - c[2]) < 2e-3 * abs(c[2]) and abs(a[3] - c[3]) < 2e-3 * abs(c[3]), (noise, a[2:4], c[2:4])
            for pa, pc in zip(a[4], c[4]):
                assert ((pa - pc).abs().mean() / pc.abs().mean()).item() < 1e-3, noise
        else:                # table-driven chained replays draw the sample indices in one call per step: other samples of the
            # same classes -> the same losses only statistically
            assert abs(a[2] - c[2]) < 5e-2 * abs(c[2]) and abs(a[3] - c[3]) < 5e-2 * abs(c[3]), (noise, a[2:4], c[2:4])


def test_generator_optimizer_state_reload_between_steps(dev):
    """the captured update holds raw pointers to the Adam moments and mirrors the step count on the device.
    optimizer_generator.load_state_dict() between two steps replaces those tensors: the step must notice and rebuild, and
    the second step must equal the one of a run without the reload."""
    import copy
    import zs3_oracle as zo
    from zs3_amd.gmmn_trainer import GMMNStep
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.modeling.gmmn import GMMNnetwork
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    seen = [c for c in range(21) if c not in (10, 14)]
    b = zo.make_synthetic_batch(3, 65, seed=5, with_label_emb=False)
    results = []
    for reload in (False, True):
        torch.manual_seed(1)
        m = DeepLab(num_classes=21, pretrained=False).to(dev).train()
        gen = GMMNnetwork(300, 300, 256, 256)
        gen.model[2].p = 0.0
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        gen = gen.to(dev).train()
        groups = [{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}]
        opt_g = Adam(gen.parameters(), lr=2e-4)
        step = GMMNStep(m, gen, SGD(groups, momentum=0.9, weight_decay=5e-4), opt_g,
                        SegmentationLosses(cuda=True).build_loss("ce"), seen=seen, unseen=[10, 14], noise="cpu")
        out = []
        for it in range(2):
            torch.manual_seed(40 + it)
            out.append(step(b["image"].to(dev), b["label"].to(dev), table=b["table"].to(dev))[:2])
            if reload and it == 0:
                old_ptr = opt_g.state[gen.model[0].weight]["exp_avg"].data_ptr()
                opt_g.load_state_dict(copy.deepcopy(opt_g.state_dict()))
                assert opt_g.state[gen.model[0].weight]["exp_avg"].data_ptr() != old_ptr
        results.append((out, [p.detach().clone() for p in gen.parameters()],
                        int(opt_g.state[gen.model[0].weight]["step"])))
    (o0, p0, s0), (o1, p1, s1) = results
    assert s0 == s1 == 2 * step.last_updates
    assert o0 == o1
    for a, c in zip(p0, p1):
        assert torch.equal(a, c)
