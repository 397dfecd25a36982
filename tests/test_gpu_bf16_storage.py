"""The 2-byte mode (BASELINE configs[4] "bf16"; ops.set_storage(torch.bfloat16)): activations and the gradients between layers are
bf16 in HBM, products plain bf16, accumulation / statistics / parameters / weight gradients fp32.

Two kinds of check.  (1) EXACTNESS of the storage change: every kernel computes in fp32 registers either way, so on inputs that
are bf16-representable the bf16-stored call must produce exactly the fp32-stored call's result rounded to bf16 (same kernel,
same accumulation order) -- bit-identical, no tolerance.  (2) ACCURACY of the mode: one layer against fp64 at the stated bf16
tolerance (2^-8 per operand and per stored tensor), and a whole training step against the fp32-storage plain-bf16 step."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


@pytest.fixture()
def bf16_mode():
    from zs3_amd import ops
    prev = ops.PREC_DEFAULT
    ops.set_storage(BF)
    yield ops
    ops.set_storage(torch.float32)
    assert ops.PREC_DEFAULT == prev      # set_storage is a round trip


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def q(t):
    """round to bf16 and back: a bf16-representable fp32 tensor"""
    return t.to(BF).float()


def test_elementwise_kernels_are_exact_in_bf16_storage(dev):
    """BatchNorm-apply (+ residual, ReLU, mask bits, dropout), column statistics, BN-backward sums and apply, max-pool, bilinear
    resize and their backwards, dropout, n-ary sum, pooled mean: bf16 in / bf16 out == round(fp32 in / fp32 out) bit for bit."""
    from zs3_amd import ops
    from zs3_amd import functional as Fz
    g = torch.Generator(device=dev).manual_seed(3)
    n, h, w, c = 2, 19, 23, 64
    x = q(torch.randn(n, h, w, c, device=dev, generator=g))
    r = q(torch.randn(n, h, w, c, device=dev, generator=g))
    sc, sh = torch.rand(c, device=dev, generator=g) + 0.5, torch.randn(c, device=dev, generator=g)
    xb, rb = x.to(BF), r.to(BF)
    m32 = torch.empty(n * h * w * c // 4, dtype=torch.uint8, device=dev)
    m16 = torch.empty_like(m32)
    a32 = ops.affine_act(x, sc, sh, res=r, act=1, mask_out=m32, drop=(0.3, 1234))
    a16 = ops.affine_act(xb, sc, sh, res=rb, act=1, mask_out=m16, drop=(0.3, 1234))
    assert a16.dtype == BF and torch.equal(a16, a32.to(BF)) and torch.equal(m16, m32)
    # the casts
    assert torch.equal(ops.cast(x, BF), xb) and torch.equal(ops.cast(xb, torch.float32), x)
    # statistics: fp32 sums of the same values
    assert torch.equal(ops.colstats(xb), ops.colstats(x))
    mean, istd = torch.randn(c, device=dev, generator=g) * 0.1, torch.rand(c, device=dev, generator=g) + 0.5
    dA = q(torch.randn(n, h, w, c, device=dev, generator=g))
    p32 = ops.bn_bwd_stats(dA, None, x, mean, istd, mask_bits=m32)
    p16 = ops.bn_bwd_stats(dA.to(BF), None, xb, mean, istd, mask_bits=m32)
    assert torch.equal(p16, p32)
    gamma, c1, c2 = torch.rand(c, device=dev, generator=g) + 0.5, torch.randn(c, device=dev, generator=g) * 0.01, torch.randn(c, device=dev, generator=g) * 0.01
    d32, r32 = torch.empty_like(x), torch.empty_like(x)
    d16, r16 = torch.empty_like(xb), torch.empty_like(xb)
    ops.bn_act_bwd(dA, None, x, mean, istd, gamma, c1, c2, dy=d32, dres=r32, mask_bits=m32)
    ops.bn_act_bwd(dA.to(BF), None, xb, mean, istd, gamma, c1, c2, dy=d16, dres=r16, mask_bits=m32)
    assert torch.equal(d16, d32.to(BF)) and torch.equal(r16, r32.to(BF))
    # pooling / resize
    o32, i32 = ops.maxpool_fwd(x)
    o16, i16 = ops.maxpool_fwd(xb)
    assert torch.equal(o16, o32.to(BF)) and torch.equal(i16, i32)
    gy = q(torch.randn(o32.shape, device=dev, generator=g))
    assert torch.equal(ops.maxpool_bwd(gy.to(BF), i16, (h, w)), ops.maxpool_bwd(gy, i32, (h, w)).to(BF))
    assert torch.equal(ops.bilinear_fwd(xb, (41, 37)), ops.bilinear_fwd(x, (41, 37)).to(BF))
    gu = q(torch.randn(n, 41, 37, c, device=dev, generator=g))
    assert torch.equal(ops.bilinear_bwd(gu.to(BF), (h, w)), ops.bilinear_bwd(gu, (h, w)).to(BF))
    assert torch.equal(ops.dropout(xb, 0.2, 99), ops.dropout(x, 0.2, 99).to(BF))
    assert torch.equal(ops.group_colsum(xb, n, 0.25), ops.group_colsum(x, n, 0.25).to(BF))
    # n-ary sum through Fz.fork: fp32 sum of the three gradients in consumer order, rounded once
    gs = [q(torch.randn(n, h, w, c, device=dev, generator=g)) for _ in range(3)]
    t = xb.clone().requires_grad_(True)
    torch.autograd.backward(list(Fz.fork(t, 3)), [v.to(BF) for v in gs])
    assert t.grad.dtype == BF and torch.equal(t.grad, ((gs[0] + gs[1]) + gs[2]).to(BF))


CASES = [  # N,H,W,Cin,Cout,k,stride,dil
    (2, 33, 33, 256, 256, 3, 1, 1), (2, 33, 33, 512, 512, 3, 1, 4), (2, 65, 65, 128, 128, 3, 2, 1), (3, 17, 19, 64, 256, 1, 1, 1),
    (2, 65, 65, 256, 512, 1, 2, 1), (2, 17, 17, 2048, 256, 3, 1, 18), (1, 67, 65, 304, 256, 3, 1, 1), (2, 9, 9, 256, 48, 1, 1, 1),
    (16, 33, 33, 1024, 256, 1, 1, 1), (16, 33, 33, 256, 1024, 1, 1, 1), (4, 65, 65, 128, 128, 3, 1, 1),
    # odd numbers of 64-channel K steps with BatchNorm sums (1, 9, 3 steps; layer 1 at its real size): the peeled last step of the
    # two-stage loop raced with the staging of the sums until the round-4 barrier (conv_igemm.hip, epilogue 1)
    (16, 129, 129, 64, 64, 1, 1, 1), (8, 129, 129, 64, 64, 3, 1, 1), (8, 129, 129, 192, 64, 1, 1, 1),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_kernels_are_exact_in_bf16_storage(dev, case):
    """forward, data gradient (with an accumulate epilogue) and weight gradient: the launch on bf16-stored tensors == the launch
    on the same values stored as fp32 with plain-bf16 products, rounded to bf16 -- whatever kernel family the rules pick for
    either; both against fp64 on the bf16-rounded operands as well (the kernels are exact for what they are asked to compute).
    Summation order differs between kernel families, so "equal" is: within one bf16 ulp of each other, and each within 3e-5 of
    fp64 before rounding."""
    from zs3_amd import ops
    from zs3_amd.functional import _pad_channels
    n, h, w, ci, co, k, s, d = case
    g = torch.Generator(device=dev).manual_seed(n * h + ci + co)
    x = q(torch.randn(n, h, w, ci, device=dev, generator=g))
    wt = q(torch.randn(co, ci, k, k, device=dev, generator=g) / (ci * k * k) ** 0.5)
    pad = d * (k // 2)
    ops.set_storage(BF)
    try:
        wp = ops.prep_weight(wt)
        y32, _ = ops.conv2d_fwd(x, wp, s, pad, d, out_dtype=torch.float32, prec=1)
        y16, st16 = ops.conv2d_fwd(x.to(BF), wp, s, pad, d, want_stats=True)
        assert y16.dtype == BF
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), stride=s, padding=pad, dilation=d).permute(0, 2, 3, 1)
        assert rel(y32, ref) < 3e-5
        assert rel(y16, y32.to(BF)) < 2 ** -7          # one bf16 ulp where the two kernels' fp32 sums straddle a rounding boundary
        assert rel(y16, ref) < 2 ** -8 + 3e-5
        # BatchNorm partial sums come from the fp32 accumulators
        ssum = st16[:, 0].double().sum(0)
        assert ((ssum - ref.sum((0, 1, 2))).abs().max() / ref.abs().sum((0, 1, 2)).max()).item() < 1e-5
        dy = q(torch.randn(y32.shape, device=dev, generator=g))
        dyp, dyp16 = _pad_channels(dy, 8), _pad_channels(dy.to(BF), 8)
        base = q(torch.randn(n, h, w, ci, device=dev, generator=g))
        dx32 = base.clone()
        ops.conv2d_dgrad(dyp, wp, (h, w), s, pad, d, out=dx32, accumulate=True, prec=1)
        dx16 = base.to(BF)
        ops.conv2d_dgrad(dyp16, wp, (h, w), s, pad, d, out=dx16, accumulate=True)
        assert rel(dx16, dx32.to(BF)) < 2 ** -7
        dw32 = ops.conv2d_wgrad(dyp, x, co, ci, k, k, s, pad, pad, d, prec=1)
        dw16 = ops.conv2d_wgrad(dyp16, x.to(BF), co, ci, k, k, s, pad, pad, d)
        dwm = ops.conv2d_wgrad(dyp, x.to(BF), co, ci, k, k, s, pad, pad, d)        # fp32 gradient x bf16 activation (the classifier's layer)
        assert dw16.dtype == torch.float32 and rel(dw16, dw32) < 3e-5 and rel(dwm, dw32) < 3e-5
    finally:
        ops.set_storage(torch.float32)
        ops.PREC_DEFAULT = 3


def _tamed(dev, seed=1):
    from zs3_amd.modeling.deeplab import DeepLab
    torch.manual_seed(seed)
    m = DeepLab(num_classes=21, pretrained=False)
    for name, mod in m.named_modules():
        if name.endswith("bn3"):
            mod.weight.data.fill_(0.1)
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m.to(dev).train()


def _train_step(dev, storage):
    from zs3_amd import ops
    from zs3_amd.utils.loss import SegmentationLosses
    from zs3_amd.utils.synthetic import make_batch
    ops.set_storage(storage)
    ops.PREC_DEFAULT = 1
    try:
        m = _tamed(dev)
        b = make_batch(4, 161, seed=5, device=dev)
        out = m(b["image"])
        loss = SegmentationLosses(cuda=True).build_loss("ce")(out, b["label"])
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
        stats = {k: v.detach().clone() for k, v in m.state_dict().items() if "running" in k}
        return out.detach().float(), float(loss.detach()), grads, stats
    finally:
        ops.set_storage(torch.float32)
        ops.PREC_DEFAULT = 3


def _stage(dev, storage, stride):
    """a residual stage as the backbone runs it -- projection block (with or without stride) + two identity blocks: pass-through
    skip tensors, the lazily masked skip gradient inside conv1's dgrad epilogue, BN-backward sums from the consumer's dgrad
    (BnLink), mask bits -- on fixed inputs, in the given storage type"""
    from zs3_amd import ops
    from zs3_amd.modeling.backbone.resnet import Bottleneck
    from zs3_amd.modeling.layers import BatchNorm2d, Conv2d, to_channels_last_
    ops.set_storage(storage)
    ops.PREC_DEFAULT = 1
    try:
        torch.manual_seed(11)
        down = torch.nn.Sequential(Conv2d(256, 512, kernel_size=1, stride=stride, bias=False), BatchNorm2d(512))
        blocks = torch.nn.Sequential(Bottleneck(256, 128, stride, 1, down), Bottleneck(512, 128, 1, 1), Bottleneck(512, 128, 1, 2))
        for name, mod in blocks.named_modules():
            if name.endswith("bn3"):
                mod.weight.data.fill_(0.3)
        blocks = to_channels_last_(blocks).to(dev).train()
        g = torch.Generator(device=dev).manual_seed(12)
        x = q(torch.randn(4, 34, 34, 256, device=dev, generator=g)).to(storage).requires_grad_(True)
        y = x
        for blk in blocks:
            y = blk.forward_nhwc(y)
        up = q(torch.randn(y.shape, device=dev, generator=g)).to(storage)
        y.backward(up)
        torch.cuda.synchronize()
        return y.detach().float(), x.grad.float(), {k: p.grad.float().clone() for k, p in blocks.named_parameters()}
    finally:
        ops.set_storage(torch.float32)
        ops.PREC_DEFAULT = 3


@pytest.mark.parametrize("stride", [1, 2])
def test_residual_stage_backward_in_bf16_storage(dev, stride):
    """The whole-network comparison below is dominated by forward noise through 101 chaotic layers; this one pins the BACKWARD
    plumbing of the 2-byte mode where it is well conditioned: three bottleneck blocks on identical inputs and an identical
    upstream gradient, bf16 storage against fp32 storage (same plain-bf16 products).  Nine convolutions and ten BatchNorms
    deep, every stored tensor rounded to 2^-9: outputs within 1e-2 (delivered 5.6e-3).  Gradients: a forward that differs by
    5e-3 flips the ReLU mask of the ~0.4 % of pre-activations that lie that close to zero, and a flipped element is an O(1) error
    in that element -- sqrt(0.004) = 6e-2 relative L2 per ReLU layer, whatever the kernels' accuracy (the same effect DESIGN.md
    section 7 records for the stem at 513x513).  Delivered: input gradient 9.4e-2, worst parameter gradient 0.13; asserted 0.2 --
    a missing skip gradient, a wrong mask or a mixed-up tensor is an O(1) error.  The kernels themselves are held bit-exact
    against their fp32-storage forms above."""
    y32, dx32, g32 = _stage(dev, torch.float32, stride)
    y16, dx16, g16 = _stage(dev, BF, stride)
    def l2(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    e_y, e_dx = l2(y16, y32), l2(dx16, dx32)
    worst = max((l2(g16[k], g32[k]), k) for k in g32)
    print(f"[bf16 storage, residual stage stride {stride}] output {e_y:.2e} input gradient {e_dx:.2e} worst parameter gradient {worst[0]:.2e} ({worst[1]})")
    assert e_y < 1e-2 and e_dx < 0.2 and worst[0] < 0.2


def test_training_step_in_bf16_storage_vs_fp32_storage(dev):
    """One supervised step (tamed network, 129x129, B=2) with bf16-stored activations and gradients against the same step with
    fp32 storage and the same plain-bf16 products.  Stated tolerance of the mode: every stored tensor carries 2^-9 relative
    rounding; through 101 train-mode layers of a randomly initialised network that is 6.5e-2 on the class scores (B = 4, 161x161).
    Class scores stay fp32 tensors; every parameter gets an fp32 gradient."""
    o32, l32, g32, s32 = _train_step(dev, torch.float32)
    o16, l16, g16, s16 = _train_step(dev, BF)
    assert o16.dtype == torch.float32 and set(g16) == set(g32)
    e_out = ((o16 - o32).norm() / o32.norm()).item()
    print(f"[bf16 storage] class scores {e_out:.2e} (relative L2), loss {l16:.5f} vs {l32:.5f}")
    assert e_out < 0.15 and abs(l16 - l32) < 1e-2 * abs(l32)      # delivered 6.5e-2 / 1e-4
    worst = 0.0
    for k in ("decoder.pred_conv.weight", "decoder.last_conv.4.weight", "decoder.last_conv.0.weight", "decoder.conv1.weight", "aspp.conv1.weight",
              "aspp.aspp1.atrous_conv.weight", "aspp.aspp4.atrous_conv.weight", "aspp.global_avg_pool.1.weight",
              "backbone.layer4.2.conv3.weight", "backbone.layer4.0.conv1.weight", "backbone.layer3.22.conv3.weight",
              "backbone.layer3.10.conv2.weight", "backbone.layer3.0.conv1.weight", "backbone.layer2.0.conv1.weight",
              "backbone.layer1.0.conv1.weight", "backbone.conv1.weight", "backbone.bn1.weight"):
        e = ((g16[k] - g32[k]).norm() / g32[k].norm().clamp_min(1e-30)).item()
        print(f"    grad {k}: {e:.2e}")
        worst = max(worst, e)
        assert g16[k].dtype == torch.float32
    # the gradients inherit the forward's 6.5e-2 (the class scores they are gradients OF differ by that much) and grow with
    # depth from there (delivered: classifier 3e-2, decoder 0.11-0.27, ASPP 0.22-0.40, backbone 0.34-0.55): a sanity bound on the
    # whole chain; test_residual_stage_backward_in_bf16_storage is the tight check of the backward kernels
    assert ((g16["decoder.pred_conv.weight"] - g32["decoder.pred_conv.weight"]).norm() / g32["decoder.pred_conv.weight"].norm()).item() < 0.1
    assert worst < 0.9
    for k in s32:
        if "num_batches" not in k:
            assert ((s16[k].double() - s32[k].double()).norm() / s32[k].double().norm().clamp_min(1e-30)).item() < 2e-2, k


@pytest.mark.parametrize("storage", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_training_step_is_bit_reproducible(dev, storage):
    """The same step on the same batch twice: logits, loss and every gradient bitwise equal, in both storage forms (no float
    atomics anywhere in the library; what broke this in the 2-byte mode was an LDS race in the BatchNorm-sum epilogue of the
    register-staged conv kernel, visible as run-to-run different losses in bench.py: tools/probe/determinism.py)."""
    from zs3_amd import ops
    from zs3_amd.utils.loss import SegmentationLosses
    from zs3_amd.utils.synthetic import make_batch
    ops.set_storage(storage)
    try:
        m = _tamed(dev)
        b = make_batch(8, 257, seed=5, device=dev)
        crit = SegmentationLosses(cuda=True).build_loss("ce")
        runs = []
        for _ in range(3):
            for p in m.parameters():
                p.grad = None
            out = m(b["image"])
            loss = crit(out, b["label"])
            loss.backward()
            torch.cuda.synchronize()
            runs.append((out.detach().clone(), loss.detach().clone(), [p.grad.detach().clone() for p in m.parameters() if p.grad is not None]))
        for out, loss, grads in runs[1:]:
            assert torch.equal(out, runs[0][0]) and torch.equal(loss, runs[0][1])
            assert not any(torch.isnan(g).any() for g in grads)
            bad = [i for i, (g, g0) in enumerate(zip(grads, runs[0][2])) if not torch.equal(g, g0)]
            assert not bad, bad[:5]
    finally:
        ops.set_storage(torch.float32)
        ops.PREC_DEFAULT = 3


@pytest.mark.gpu
def test_generator_rows_stay_fp32_in_the_2_byte_mode():
    """Round 6: the eager generator path of the GMMN / GCN-context step (images with an unseen class, train_pascal_GMMN.py:242) runs
    its two Linear layers as row GEMMs; in the 2-byte mode the launcher's default output type is bf16, and the fp32-only row kernels
    behind it (zs3_scatter_rows ...) then read a bf16 buffer as fp32 -- garbage features and, once in a few runs, a memory fault.
    The rows keep their own type now, and the row kernels refuse anything but fp32."""
    from zs3_amd import ops
    from zs3_amd.gmmn_trainer import _rows_gemm
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    x = torch.randn(144, 600, generator=g).to(dev)
    w = (torch.randn(256, 600, generator=g) / 600 ** 0.5).to(dev)
    b = torch.randn(256, generator=g).to(dev)
    ref = (x.double() @ w.double().t() + b.double()).float()
    idx = torch.randperm(289, generator=g)[:144].to(dev)
    outs = {}
    for storage in (torch.float32, torch.bfloat16):
        ops.set_storage(storage)
        try:
            y = _rows_gemm(x, ops.prep_weight(w), b)
            assert y.dtype == torch.float32 and tuple(y.shape) == (144, 256)
            rows = torch.zeros(289, 256, device=dev)
            ops.scatter_rows(y, idx, rows)
            torch.cuda.synchronize()
            assert torch.equal(rows[idx], y)
            outs[storage] = y.clone()
            with pytest.raises(TypeError):
                ops.scatter_rows(y.bfloat16(), idx, rows)
            with pytest.raises(TypeError):
                ops.gather_rows(rows.bfloat16(), idx)
        finally:
            ops.set_storage(torch.float32)
    assert ((outs[torch.float32] - ref).abs().max() / ref.abs().max()).item() < 1e-4           # x3 products
    assert ((outs[torch.bfloat16] - ref).abs().max() / ref.abs().max()).item() < 2e-2          # plain bf16 products, fp32 rows
