"""Thin, non-differentiable wrappers over the C ABI (one Python function per kernel family).

Internal activation layout is NHWC: a tensor [N, H, W, C] whose last dim is contiguous and whose
pixel stride (`ld`) may exceed C (channel-slice views of a wider buffer).  Logical-NCHW tensors with
channels_last strides are the same memory: `nhwc(t)` / `nchw(t)` convert without copying.
"""
import ctypes
import os
from dataclasses import dataclass

import torch

from ._lib import F, I, P, check, lib, require_gpu, stream

PREC_DEFAULT = 3  # bf16x3 split; 1 = plain bf16 inputs; 0 = exact fp32 (test mode, set_exact_fp32)
PROFILE = None    # set to a list to record (tag, algorithmic_flops, start_event, end_event, tile_cfg) per conv launch
PROFILE_CFGS = None   # optional set of tile_cfg values to restrict the recording to (event pairs serialise kernel boundaries)
PROFILE_SAMPLE = None  # optional [stride, phase, counter]: record every stride-th eligible launch (an event pair costs host time)
PROFILE_GEOM = None    # optional list: one (m, ncols, k, taps, stride, dil, dgrad, epilogue kind, io) per PROFILE record (tools/probe/step_layers.py)


WGRAD_STRIP = os.environ.get("ZS3_WGRAD_STRIP", "1") == "1"   # strip-resident weight gradient of the 3x3 stride-1 layers
PW = os.environ.get("ZS3_PW", "1") == "1"                      # persistent pointwise kernel for the 1x1 stride-1 layers
PW_FORCE = 0        # 51 / 52: every eligible 1x1 layer on that tile (set by A/B probes; the rule is pick_pw_tile)
WGRAD_PW = os.environ.get("ZS3_WGRAD_PW", "1") == "1"         # producer-split weight gradient of the 1x1 stride-1 layers
HALO = os.environ.get("ZS3_HALO", "1") == "1"     # strip-resident kernel (tile_cfg 41 / 42) for the multi-tap stride-1 layers


ACT_DTYPE = torch.float32   # element type of activation tensors in HBM: torch.float32, or torch.bfloat16 = the 2-byte mode (set_storage)
BF16 = torch.bfloat16


def _io(inp, out=None):
    """the `io` argument of the C ABI from the element types of a call's activation input / output tensors"""
    return (1 if inp is not None and inp.dtype == BF16 else 0) | (2 if out is not None and out.dtype == BF16 else 0)


def _same_type(ref, *others):
    for t in others:
        if t is not None and t.dtype != ref.dtype:
            raise TypeError(f"activation tensors of one call must share an element type: {t.dtype} next to {ref.dtype}")


_fp32_state = None   # (PREC_DEFAULT, weight-gradient kernel selection) in force when the 2-byte mode was entered


def set_storage(dtype):
    """Element type of the activation tensors the fused layers allocate: torch.float32 (BASELINE configs[1-3]) or torch.bfloat16
    -- the 2-byte mode of configs[4]: conv outputs, BatchNorm-applied activations and every gradient between layers are
    bf16 in HBM (half the bytes of every HBM-bound pass and of every conv operand), products are plain bf16 (prec = 1),
    accumulation / statistics / parameters / weight gradients stay fp32.  Image input and class scores stay fp32.
    set_storage(torch.float32) restores the arithmetic (PREC_DEFAULT) and the weight-gradient kernel selection that were in
    force when the 2-byte mode was entered: a round trip is the identity."""
    global ACT_DTYPE, PREC_DEFAULT, _fp32_state
    from . import functional as Fz
    if dtype not in (torch.float32, BF16):
        raise ValueError("activation storage is torch.float32 or torch.bfloat16")
    if dtype == BF16 and ACT_DTYPE != BF16:
        # (zs3_conv_wgrad_set_kernel returns the previous selection: ask by setting, the value is overwritten just below)
        _fp32_state = (PREC_DEFAULT, int(lib().zs3_conv_wgrad_set_kernel(I(1))))
        PREC_DEFAULT = 1
        lib().zs3_conv_wgrad_set_kernel(I(1))      # the LDS-DMA weight-gradient kernel moves raw fp32 rows
    elif dtype == torch.float32 and ACT_DTYPE == BF16:
        prec, wk = _fp32_state if _fp32_state is not None else (3, 0)
        PREC_DEFAULT = prec
        lib().zs3_conv_wgrad_set_kernel(I(wk))
        _fp32_state = None
    ACT_DTYPE = dtype
    Fz.PLAN_EPOCH[0] += 1
    _TILE_CHOICE.clear()
    _WGRAD_PLAN.clear()
    _MTILES.clear()
    Fz._planes.clear()          # forward planes are fp16 hi/lo in fp32 storage (f16x3), bf16 in the 2-byte mode
    Fz._refresh_tables.clear()
    Fz._defer_choice.clear()
    Fz._in_affine_choice.clear()


_exact_state = None   # (weight-gradient kernel selection, PREC_DEFAULT, HALO, PW, WGRAD_STRIP, WGRAD_PW) in force when the exact-fp32 test mode was entered


def set_exact_fp32(on=True):
    """Test mode: every convolution / linear product on v_mfma_f32_32x32x2_f32 (prec = 0 of the register-staged kernels, ~1/16 of
    the bf16 rate) -- the arithmetic of the reference's own fp32 convolutions, summation order aside.  It answers one question:
    is a difference from the reference bf16x3 arithmetic or structure?  (tests/test_gpu_model.py runs the reference's default-init
    train-mode goldens through it.)  Switches the producer-converting kernel families off, since they exist in bf16 only."""
    global PREC_DEFAULT, HALO, PW, WGRAD_STRIP, WGRAD_PW, _exact_state
    from . import functional as Fz
    if on:
        if _exact_state is None:   # (PREC_DEFAULT and the kernel-family switches in force when the mode was entered: a round trip
            # is the identity also INSIDE the 2-byte mode, where PREC_DEFAULT is 1 -- like set_storage's own save / restore)
            _exact_state = (int(lib().zs3_conv_wgrad_set_kernel(I(1))), PREC_DEFAULT, HALO, PW, WGRAD_STRIP, WGRAD_PW)
        PREC_DEFAULT, HALO, PW, WGRAD_STRIP, WGRAD_PW = 0, False, False, False, False
        lib().zs3_conv_wgrad_set_kernel(I(1))
    elif _exact_state is not None:
        wk, PREC_DEFAULT, HALO, PW, WGRAD_STRIP, WGRAD_PW = _exact_state
        lib().zs3_conv_wgrad_set_kernel(I(wk))
        _exact_state = None
    else:
        return
    Fz.PLAN_EPOCH[0] += 1
    _TILE_CHOICE.clear()
    _WGRAD_PLAN.clear()
    _MTILES.clear()
    Fz._planes.clear()
    Fz._defer_choice.clear()
    Fz._in_affine_choice.clear()
    Fz._refresh_tables.clear()


def halo_ok(x_shape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, dil, dgrad, prec, tile_cfg):
    """zs3_conv_halo_ok: can tile_cfg 41 / 42 (csrc/conv_halo.hip) run this launch?  0 = no, else the NPG template argument
    of the conv_halo_kernel<PREC, BM, NPG> instantiation it will run (the name a profiler lists the launch under)."""
    n, h, w_ = x_shape[:3]
    return int(lib().zs3_conv_halo_ok(I(n), I(h), I(w_), I(ho), I(wo), I(cin_pad), I(cin_valid), I(ldx), I(kh), I(kw), I(stride),
                                       I(pad_h), I(pad_w), I(dil), I(int(dgrad)), I(prec), I(tile_cfg)))


# Tile height of the strip-resident kernel.  Measured inside the training step (same-box A/B, tools/probe/ab_env.sh, ms per
# step): no strip kernel 48.34 / 48.40, 192-row tiles wherever they need fewer rounds 48.31 / 48.07, 256 everywhere 47.79 /
# 47.86, 192 everywhere 48.46 / 48.33, the rule below 47.66 / 47.68.  In isolation 192-row tiles win (182 tiles on 256 CUs
# instead of 138: 78 us against 91 us for the 3x3 256->256 layer, 99 us on the LDS-DMA kernel), but in backward the
# weight-gradient streams fill whatever CUs a launch leaves idle, so what counts there is CU-time per tile, and a 256-row
# tile spends 75 % of its K loop issuing MFMAs against 67 % for 192 rows.
HALO_BM = os.environ.get("ZS3_HALO_BM", "bwd256")   # auto | 256 | 192 | bwd256 (dgrad launches on 256-row tiles)


# Launches with fewer 256 x 128 tiles than this leave the 256-row rules (round 5, the B = 8 shard of BASELINE configs[3]: M = 8 712 at
# 33 x 33 is 35 x 2 = 70 such tiles on 256 CUs -- there 64 x 64 tiles beat the LDS-DMA kernel on the long-K 1x1 layers, 35 against 44 us,
# and 192-row strips the 256-row ones in the data gradients, 66 against 78 us; at B = 16 the same launches are 138 tiles and stay)
SMALL_LAUNCH_TILES = 100


def pick_halo_tile(m, ncols, dgrad=False):
    """256- or 192-row tiles for the strip-resident kernel: the tile height that needs fewer rounds x rows on 256 CUs
    (16 x 33 x 33 output pixels x 256 channels: 138 tiles of 256 rows -> one round at 54 % of the chip; 182 tiles of
    192 rows -> one round at 71 %, each 0.75x as long)."""
    if HALO_BM in ("256", "192"):
        return 41 if HALO_BM == "256" else 42
    if HALO_BM == "bwd256" and dgrad and ((m + 255) // 256) * ((ncols + 127) // 128) >= SMALL_LAUNCH_TILES:
        return 41
    nt = (ncols + 127) // 128
    def cost(bm):
        tiles = ((m + bm - 1) // bm) * nt
        return ((tiles + 255) // 256) * (bm + 24)      # + ~24 rows' worth of prologue / epilogue per tile
    return 42 if cost(192) < cost(256) else 41


DMA_RULE = True     # the LDS-DMA kernel (tile_cfg 31) for the long-K wide layers (False: the 128x128 register-staged kernel; slower on every such layer)


def pick_tile(m, ncols, k=0, taps=1):
    """Tile / kernel choice for the implicit-GEMM conv (zs3_conv_igemm tile_cfg): 1x = register-staged 4-wave kernel
    (11: 128x128, 14: 64x64 block tile), 31 = wave-specialised 256x128 kernel fed by LDS-DMA."""
    # measured on MI355X over the 28 layer shapes of the network, forward and dgrad (tools/probe/conv_bench.py).
    # The LDS-DMA kernel wins wherever there is enough K per tile to amortise its 3-stage ring (K >= 512) and at least
    # two 128-wide column tiles; short-K / narrow layers keep the small-tile kernels: 128x128 when the launch has
    # >= ~1000 tiles (>= 2 resident per CU for several rounds), else 64x64 (4 blocks per CU).  The rules were re-checked
    # inside the whole step (same-box A/B runs of bench.py): layers that are a toss-up in isolation favour the small
    # tiles there (3x3 128->128: -0.24 ms per step on 64x64 tiles although the DMA kernel is level in isolation), and
    # the DMA kernel must keep the 1x1 1024->256 layers (+1.1 ms per step without it).  Re-checked at the end of round 2 with the
    # wgrad streams at 96 CUs (tools/probe/r2v.sh, same box, ms per step): these rules 47.0 / 47.3; short-K wide layers on the
    # DMA kernel 47.4 / 47.6, on 128x128 tiles 47.0 / 47.3; the 128x128-vs-64x64 switch at 400 tiles 47.4 / 47.1, at 3000
    # tiles 47.3 / 47.1 -- nothing left in the rules.
    if ncols <= 64:
        return 14
    if m >= 8192 and k >= 512 and ncols >= 256 and (taps > 1 or ((m + 255) // 256) * ((ncols + 127) // 128) >= SMALL_LAUNCH_TILES):
        return 31 if DMA_RULE else 11    # (multi-tap layers: the strip-resident kernel is asked next, _choose_tile)
    if ncols >= 256 and 128 <= k <= 256:
        # 1x1 layers with a short K and many column tiles that the persistent pointwise kernel does not take (the loading epilogues:
        # layer 3's conv1 data gradient 256 -> 1024 @33^2 with residual + BatchNorm-backward sums, 22 launches per step, and the
        # 128 -> 512 @65^2 ones).  Round 2 put them on 64 x 64 tiles (4-9 % faster per layer alone, 52.2 -> 51.7 ms per step then);
        # re-checked inside the round-6 step (tools/probe/r6t.sh, same box, interleaved): 128 x 128 tiles 42.37 / 42.39 ms against
        # 42.62 / 42.63 -- the round-4 epilogue (operands of four rows fetched ahead) amortises over the larger tile
        return SHORTK_TILE
    return 11 if ((m + 127) // 128) * ((ncols + 127) // 128) >= TILE_SWITCH else 14


SHORTK_TILE, TILE_SWITCH = 11, 1000
PW_MAXK = 512      # longest reduction the persistent pointwise kernel takes (longer: the LDS-DMA kernel)


def pick_pw_tile(m, ncols, k):
    """Layers that go to the persistent pointwise kernel (tile_cfg 51: 256-row tiles, 52: 128-row tiles; csrc/conv_pw.hip),
    0 = none (pick_tile's kernels)."""
    if PW_FORCE:
        return PW_FORCE
    # forward table at B=16 (tools/probe/r3m.sh; rules' kernel / 51 / 52, us): 64->256 @129^2 125 / 86 / 80, 128->512 @65^2 75 / 68 / 62,
    # 256->1024 @33^2 62 / 63 / 52, 512->2048 @33^2 159 / 142 / 144 -- the write-heavy short reductions, where the stores
    # draining under the next tile pay; with K >= 1024 (1024->256: 52 / 60 / 64) the LDS-DMA kernel stays ahead
    if ncols >= 256 and k <= PW_MAXK and m >= 8192:
        return 52
    return 0


def pw_ok(xshape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, tile_cfg):
    n, h, w_, _ = xshape
    return bool(lib().zs3_conv_pw_ok(I(n), I(h), I(w_), I(ho), I(wo), I(cin_pad), I(cin_valid), I(ldx), I(kh), I(kw),
                                     I(stride), I(pad_h), I(pad_w), I(tile_cfg)))


_zero_pages = {}


def zero_page(device):
    """256 zero floats per device: the address masked kernel loads are redirected to."""
    key = (device.type, device.index)
    if key not in _zero_pages:
        _zero_pages[key] = torch.zeros(256, dtype=torch.float32, device=device)
    return _zero_pages[key]


def _round_up(a, b):
    return (a + b - 1) // b * b


def nhwc(t):
    """Logical NCHW (channels_last memory) -> [N,H,W,C] view; copies only if the memory is not NHWC."""
    v = t.permute(0, 2, 3, 1)
    try:
        if v.stride(3) != 1 and v.shape[3] > 1:
            raise AssertionError
        _check_nhwc(v)
    except AssertionError:
        v = v.contiguous()
    return v


def nchw(t):
    return t.permute(0, 3, 1, 2)


def cast(x, dtype):
    """fp32 <-> bf16 copy of an activation tensor (zs3_affine_act as a cast); x itself when it already has the type"""
    if x.dtype == dtype:
        return x
    return affine_act(x, out_dtype=dtype)


def _check_nhwc(t):
    if t.dim() == 4 and t.is_contiguous():      # (the common case; ~2000 calls per training step: one C call instead of ten)
        return t.shape[3]
    assert t.dim() == 4 and (t.stride(3) == 1 or t.shape[3] == 1), "expected an NHWC tensor with contiguous channels"
    ld = t.stride(2) if t.shape[2] > 1 else (t.stride(1) if t.shape[1] > 1 else (t.stride(0) if t.shape[0] > 1 else t.shape[3]))
    # rows must be densely packed pixels: stride(1) == W*ld, stride(0) == H*W*ld
    if t.shape[2] > 1 and t.shape[1] > 1:
        assert t.stride(1) == t.shape[2] * ld, "NHWC rows must be densely packed"
    if t.shape[0] > 1 and t.shape[1] * t.shape[2] > 1:
        assert t.stride(0) == t.shape[1] * t.shape[2] * ld, "NHWC images must be densely packed"
    return ld


@dataclass
class WeightPlanes:
    f_pk: torch.Tensor   # forward operand  [cout][K/32][2][32] bf16
    t_pk: torch.Tensor   # dgrad operand    [cin][K'/32][2][32] bf16 (or None)
    cout: int
    cin: int
    kh: int
    kw: int
    cin_pad: int
    cout_pad: int
    f_fmt: int = 0       # 0: the forward plane holds bf16 hi/lo (prec 3 / 1), 1: fp16 hi/lo (prec 4, "f16x3"), 2: fp32 (prec 0)


# Forward convolutions of the fp32-storage network on the fp16 hi/lo split (prec = 4) instead of the bf16 one: same three MFMAs
# at the same rate, 2^-22-class products.  Measured on the reference's default-init train-mode goldens: logits 2.0e-3 from the
# goldens where bf16x3 sits at 3.6e-2 (tools/probe/split_emulation.py predicted it on the CPU, tests/test_gpu_model.py holds it).
# Data- and weight-gradient launches stay bf16x3: gradients need fp32's exponent range.  ZS3_FWD_F16=0 restores bf16x3 forward.
FWD_F16 = os.environ.get("ZS3_FWD_F16", "1") == "1"


def fwd_f16():
    """whether forward planes are being prepared as fp16 hi/lo (fp32 storage, bf16x3 default arithmetic, switch on)"""
    return FWD_F16 and PREC_DEFAULT == 3 and ACT_DTYPE == torch.float32


def prep_weight(w, need_t=True, cin_pad=None, f16_forward=False):
    """w: [Cout, Cin, KH, KW] parameter (any strides) or [Cout, Cin] linear weight -> bf16 hi/lo planes (f16_forward: the forward
    plane as fp16 hi/lo for prec = 4 launches; the data-gradient plane stays bf16 hi/lo)."""
    require_gpu(w)
    if w.dtype != torch.float32:
        raise TypeError(f"weights are fp32 (master copies): got {w.dtype}")
    if w.dim() == 2:
        w = w[:, :, None, None]
    cout, cin, kh, kw = w.shape
    wl = w.detach().permute(0, 2, 3, 1).contiguous()  # [Cout][KH][KW][Cin]; a no-op view for channels_last params
    cin_pad = cin_pad or _round_up(cin, 32)
    cout_pad = _round_up(cout, 32)
    taps = kh * kw
    dev = w.device
    f_pk = torch.empty((cout, 2 * taps * cin_pad), dtype=torch.bfloat16, device=dev)
    t_pk = torch.empty((cin, 2 * taps * cout_pad), dtype=torch.bfloat16, device=dev) if need_t else None
    f_fmt = 2 if PREC_DEFAULT == 0 else (1 if f16_forward else 0)
    prep = (lib().zs3_prep_weight, lib().zs3_prep_weight_f16fwd, lib().zs3_prep_weight_f32)[f_fmt]   # same buffers: 4 bytes per element in every form
    check(prep(P(wl), P(f_pk), P(t_pk), I(cout), I(taps), I(cin), I(cin_pad), I(cout_pad), stream()), "zs3_prep_weight")
    return WeightPlanes(f_pk, t_pk, cout, cin, kh, kw, cin_pad, cout_pad, f_fmt)


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


_TILE_CHOICE, _MTILES = {}, {}
TILE_OVERRIDE = {}     # (m, ncols, cin, taps, stride, dgrad, store-only epilogue, io) -> tile_cfg, set by A/B probes before the first launch


def _choose_tile(tile_cfg, xshape, m, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, dil, ncols, dgrad, prec,
                 pw_epilogue, io=0):
    """tile_cfg of a conv launch: the caller's explicit choice where that kernel can run the launch, else the rules."""
    if TILE_OVERRIDE and tile_cfg == 0:     # probe: one geometry on another kernel (tools/probe/r6q.sh)
        tile_cfg = TILE_OVERRIDE.get((m, ncols, min(cin_pad, cin_valid), kh * kw, stride, int(bool(dgrad)), int(bool(pw_epilogue)), io), 0)
    if io:
        return _choose_tile16(tile_cfg, xshape, m, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, dil, ncols, dgrad,
                              prec, pw_epilogue, io)
    if prec == 0:   # exact fp32 (test mode): register-staged kernel only
        return tile_cfg if 0 < tile_cfg <= 14 else (14 if ncols <= 64 or ((m + 127) // 128) * ((ncols + 127) // 128) < 1000 else 11)
    if tile_cfg in (51, 52) and not (pw_epilogue and pw_ok(xshape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, tile_cfg)):
        tile_cfg = 0               # not a 1x1 stride-1 layer, or a loading epilogue: the persistent kernel leaves those to the others
    if tile_cfg == 0 and PW and kh * kw == 1 and pw_epilogue:
        cand = pick_pw_tile(m, ncols, min(cin_pad, cin_valid))
        if cand and pw_ok(xshape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, cand):
            tile_cfg = cand
    if tile_cfg == 0:
        tile_cfg = pick_tile(m, ncols, kh * kw * min(cin_pad, cin_valid), kh * kw)
        if HALO and tile_cfg == 31 and kh * kw > 1:
            cand = pick_halo_tile(m, ncols, dgrad)
            if halo_ok(xshape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, dil, dgrad, prec, cand):
                tile_cfg = cand
            elif cand == 41 and halo_ok(xshape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, dil, dgrad,
                                        prec, 42):
                tile_cfg = 42      # the strip of a 256-row tile does not fit the LDS (ASPP, dilation 12): 192 rows do
    elif tile_cfg in (41, 42) and not halo_ok(xshape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, dil,
                                              dgrad, prec, tile_cfg):
        tile_cfg = 31              # not a stride-1 same-size multi-tap layer (or the strip does not fit)
    return tile_cfg


# persistent pointwise kernel on bf16-stored tensors (A16 producers, DPP-packed bf16 stores): correct and OFF -- with 64-channel K
# steps the register-staged kernel is the faster one for plain-bf16 1x1 layers (same-box A/B: 32.80 / 32.81 ms per step with the
# persistent kernel, 32.25 / 32.23 without; tools/probe/r4i.sh): eight MFMAs per wave and barrier leave its K loop latency-bound
PW16 = False
HALO16 = True     # strip-resident kernel on bf16-stored tensors


def _choose_tile16(tile_cfg, xshape, m, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, dil, ncols, dgrad, prec,
                   pw_epilogue, io):
    """Kernel choice when x (io bit 0) and / or y (bit 1) are bf16 in memory.  The LDS-DMA kernel (31) moves raw fp32 rows and is
    out; the strip-resident kernel reads bf16 strips (16 bytes = 8 channels per lane), the persistent pointwise kernel bf16 rows,
    and the register-staged kernel takes everything else."""
    x16 = bool(io & 1)
    vec8 = ldx % 8 == 0 and cin_valid % 8 == 0
    # 1x1 launches on the persistent kernel: store-only epilogues
    pw_able = io == 3 and vec8 and ncols % 2 == 0 and bool(pw_epilogue)   # (both sides bf16: the kernel has one element type per instantiation)
    if tile_cfg in (41, 42) and not (kh * kw == 9 and (not x16 or (vec8 and prec == 1)) and halo_ok(
            xshape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, dil, dgrad, prec, tile_cfg)):
        tile_cfg = 0
    if tile_cfg in (51, 52) and not (pw_able and pw_ok(xshape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, tile_cfg)):
        tile_cfg = 0
    if tile_cfg == 31:
        tile_cfg = 0
    if tile_cfg:
        return tile_cfg
    if HALO16 and kh * kw == 9 and m >= 8192 and ncols >= 128 and (not x16 or vec8) and (prec == 1 or not x16):
        cand = pick_halo_tile(m, ncols, dgrad)
        for c in (cand, 42 if cand == 41 else 41):
            if halo_ok(xshape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, dil, dgrad, prec, c):
                return c
    if PW16 and pw_able and kh * kw == 1 and m >= 8192 and ncols >= 128 and \
            pw_ok(xshape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, 52):
        return 52
    if ncols <= 64:
        return 14
    return 11 if ((m + 127) // 128) * ((ncols + 127) // 128) >= TILE16_SWITCH else 14


# 2-byte mode: 128 x 128 tiles from this many of them, 64 x 64 below.  512 through round 5; inside the round-6 step (r6t.sh): always
# 128 x 128 27.16 / 27.16 ms, switch at 512: 27.38 / 27.46, at 1200: 27.43 / 27.54, at 3000: 27.96 / 27.94, always 64 x 64: 27.95 / 28.04
TILE16_SWITCH = 0


def conv_igemm(x, w_pk, *, ho, wo, cin_pad, cin_valid, kh, kw, stride, pad_h, pad_w, dil, ncols, out=None,
               scale=None, shift=None, res=None, want_stats=False, act=0, leak=0.2, accumulate=False, dgrad=False,
               prec=None, tile_cfg=0, bn_bwd=None, res_mask_bits=None, in_affine=None, out_dtype=None):
    """Raw launcher.  x: NHWC [N,H,W,*]; returns (y [N,ho,wo,ncols] or `out`, stat_partial or None).
    bn_bwd = (y, mean, invstd, mask_scale, mask_shift, mask_bits): also return the BatchNorm-backward partial sums
    (sum dz, sum dz*xhat per row tile) of the layer the output gradient belongs to (zs3_conv_igemm_bnstats).
    res_mask_bits: `res` is added through a ReLU mask given as sign bytes (the residual block's skip gradient).
    in_affine = (scale, shift): x is read through max(x * scale + shift, 0) by the kernel's producer waves (the BatchNorm-apply +
    ReLU of the layer that produced x; only the strip-resident / persistent pointwise kernels: ask `consumer_applies_bn` first)."""
    require_gpu(x, w_pk, out, scale, shift, res)
    prec = PREC_DEFAULT if (prec is None or PREC_DEFAULT == 0) else prec   # (the exact-fp32 test mode overrides per-call choices: its planes are fp32)
    n, h, w_, _ = x.shape
    ldx = _check_nhwc(x)
    if out is None:
        out = torch.empty((n, ho, wo, ncols), dtype=out_dtype or ACT_DTYPE, device=x.device)
    ldy = _check_nhwc(out)
    ldr = _check_nhwc(res) if res is not None else 0
    io = _io(x, out)
    _same_type(out, res, bn_bwd[0] if bn_bwd is not None else None)
    if io & 1 and prec != 1:
        raise ValueError("a bf16-stored input needs plain-bf16 products (prec = 1)")
    m = n * ho * wo
    # store-only epilogue (no per-element loads): what the persistent pointwise kernel runs
    pw_epilogue = 1 if (res is None and not accumulate and bn_bwd is None and res_mask_bits is None) else 0
    # the kernel / tile choice depends on the launch geometry only: decided once per distinct launch (a training step repeats
    # ~60 geometries 230 times; the eligibility questions below are C calls)
    if tile_cfg in (141, 142):      # round-3 spelling of "tile_cfg 41 / 42 on a bf16-stored input"
        tile_cfg -= 100
    key = (tile_cfg, n, h, w_, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, dil, ncols, dgrad, prec, pw_epilogue,
           HALO, HALO_BM, PW, PW_FORCE, DMA_RULE, io, PW16, HALO16, PW_MAXK, SMALL_LAUNCH_TILES)
    cached = _TILE_CHOICE.get(key)
    if cached is not None:
        tile_cfg = cached
    else:
        tile_cfg = _TILE_CHOICE[key] = _choose_tile(tile_cfg, x.shape, m, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h,
                                                    pad_w, dil, ncols, dgrad, prec, pw_epilogue, io)
    stat = None
    if want_stats or bn_bwd is not None:
        mkey = (m, ncols, tile_cfg)
        mt = _MTILES.get(mkey)
        if mt is None:
            mt = _MTILES[mkey] = lib().zs3_conv_igemm_mtiles(I(m), I(ncols), I(tile_cfg))
        stat = torch.empty((mt, 2, ncols), dtype=torch.float32, device=x.device)
    prof = PROFILE is not None and (PROFILE_CFGS is None or tile_cfg in PROFILE_CFGS)
    if prof and PROFILE_SAMPLE is not None:
        PROFILE_SAMPLE[2] += 1
        prof = PROFILE_SAMPLE[2] % PROFILE_SAMPLE[0] == PROFILE_SAMPLE[1]
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if in_affine is not None:
        assert bn_bwd is None and res_mask_bits is None
        if tile_cfg not in (41, 42, 51, 52):
            raise ValueError(f"in_affine needs a producer-converting kernel; this launch runs on tile_cfg {tile_cfg}")
        check(lib().zs3_conv_igemm_in(P(x), P(w_pk), P(out), P(scale), P(shift), P(res), P(stat), I(n), I(h), I(w_),
                                      I(ho), I(wo), I(cin_pad), I(cin_valid), I(ldx), I(kh), I(kw), I(stride), I(pad_h),
                                      I(pad_w), I(dil), I(ncols), I(ldy), I(ldr), I(act), F(leak), I(int(accumulate)),
                                      I(int(dgrad)), I(prec), I(tile_cfg), P(zero_page(x.device)), P(in_affine[0]),
                                      P(in_affine[1]), I(io), stream()), "zs3_conv_igemm_in")
    elif bn_bwd is not None or res_mask_bits is not None:
        assert scale is None and shift is None and act == 0 and not want_stats
        by, bmean, bistd, bmsc, bmsh, bbits = bn_bwd if bn_bwd is not None else (None,) * 6
        require_gpu(by, bmean, bistd, bmsc, bmsh, bbits, res_mask_bits)
        check(lib().zs3_conv_igemm_bnstats(P(x), P(w_pk), P(out), P(res), P(res_mask_bits), I(n), I(h), I(w_), I(ho), I(wo),
                                           I(cin_pad),
                                           I(cin_valid), I(ldx), I(kh), I(kw), I(stride), I(pad_h), I(pad_w), I(dil),
                                           I(ncols), I(ldy), I(ldr), I(int(accumulate)), I(int(dgrad)), I(prec),
                                           I(tile_cfg), P(zero_page(x.device)), P(by),
                                           I(_check_nhwc(by) if by is not None else 0), P(bmean),
                                           P(bistd), P(bmsc), P(bmsh), P(bbits), P(stat), I(io), stream()),
              "zs3_conv_igemm_bnstats")
    else:
        check(lib().zs3_conv_igemm(P(x), P(w_pk), P(out), P(scale), P(shift), P(res), P(stat), I(n), I(h), I(w_),
                                   I(ho), I(wo), I(cin_pad), I(cin_valid), I(ldx), I(kh), I(kw), I(stride), I(pad_h),
                                   I(pad_w), I(dil), I(ncols), I(ldy), I(ldr), I(act), F(leak), I(int(accumulate)),
                                   I(int(dgrad)), I(prec), I(tile_cfg), P(zero_page(x.device)), I(io), stream()),
              "zs3_conv_igemm")
    if prof:
        e1.record()
        PROFILE.append(("conv_halo_kernel<%d, %d, %d%s>" % (prec, 256 if tile_cfg == 41 else 192, halo_ok(
                            x.shape, ho, wo, cin_pad, cin_valid, ldx, kh, kw, stride, pad_h, pad_w, dil, dgrad, prec, tile_cfg),
                            ", bf16 in" if io & 1 else "")
                        if tile_cfg in (41, 42) else
                        "conv_pw_kernel<%d, %d>" % (prec, 256 if tile_cfg == 51 else 128) if tile_cfg in (51, 52) else
                        "conv_igemm_dma<256,128,%d>" % prec if tile_cfg == 31 else f"conv_igemm<{('128,128', '128,64', '64,128', '64,64')[tile_cfg % 10 - 1]},{prec},pipe{1 + tile_cfg // 10}>",
                        2.0 * m * ncols * kh * kw * min(cin_pad, cin_valid), e0, e1, tile_cfg))
        if PROFILE_GEOM is not None:
            PROFILE_GEOM.append((m, ncols, min(cin_pad, cin_valid), kh * kw, stride, dil, int(dgrad),
                                 ("stats" if want_stats else "store") if pw_epilogue == 1 else
                                 "+".join(n for n, on in (("res", res is not None), ("acc", accumulate), ("bnbwd", bn_bwd is not None),
                                                          ("aff", scale is not None or shift is not None)) if on), io))
    return out, stat


def conv2d_fwd(x, wp, stride=1, pad=0, dil=1, **kw):
    """x: NHWC [N,H,W,C>=wp.cin] (pad channels, if any, must be zero)."""
    n, h, w_, c = x.shape
    ho = conv_out_size(h, wp.kh, stride, pad, dil)
    wo = conv_out_size(w_, wp.kw, stride, pad, dil)
    cin_valid = min(_round_up(wp.cin, 4), _check_nhwc(x))
    if wp.f_fmt == 1:        # the plane is fp16 hi/lo: the launch must multiply in fp16
        if kw.get("prec") not in (None, 3, 4) or PREC_DEFAULT != 3:
            raise ValueError("these weight planes were prepared for f16x3 forward launches (prec 4)")
        kw["prec"] = 4
    return conv_igemm(x, wp.f_pk, ho=ho, wo=wo, cin_pad=wp.cin_pad, cin_valid=cin_valid, kh=wp.kh, kw=wp.kw,
                      stride=stride, pad_h=pad, pad_w=pad, dil=dil, ncols=wp.cout, **kw)


def conv2d_dgrad(dy, wp, in_hw, stride=1, pad=0, dil=1, **kw):
    """dy: NHWC [N,Ho,Wo,C>=wp.cout] (channels beyond cout zero) -> dx [N,H,W,wp.cin]."""
    h, w_ = in_hw
    cin_valid = min(_round_up(wp.cout, 4), _check_nhwc(dy))
    out, part = conv_igemm(dy, wp.t_pk, ho=h, wo=w_, cin_pad=wp.cout_pad, cin_valid=cin_valid, kh=wp.kh,
                           kw=wp.kw, stride=stride, pad_h=pad, pad_w=pad, dil=dil, ncols=wp.cin, dgrad=True, **kw)
    return (out, part) if kw.get("bn_bwd") is not None else out


_WGRAD_PLAN = {}


WGRAD16_FAST = True   # strip-resident / pointwise weight-gradient kernels on bf16-stored operands (both bf16)


def _wgrad_plan(n, h, w_, ho, wo, kh, kw, stride, pad_h, pad_w, dil, cout, cin, io=0):
    """(kernel kind, split-K workspace floats) of a weight-gradient launch: a function of the geometry, asked of the library once
    per distinct layer.  kind: "strip" (3x3 strip-resident), "pw" (pointwise), "gemm" (the round-2 kernels)."""
    fast = io == 0 or (WGRAD16_FAST and io == 3)
    key = (n, h, w_, ho, wo, kh, kw, stride, pad_h, pad_w, dil, cout, cin, WGRAD_STRIP, WGRAD_PW, io, fast, ACT_DTYPE)
    plan = _WGRAD_PLAN.get(key)
    if plan is None:
        splitk, ws = ctypes.c_int(0), ctypes.c_long(0)
        if not fast:
            lib().zs3_conv_wgrad_plan(I(n * ho * wo), I(wo), I(cout), I(cin), I(kh * kw), ctypes.byref(splitk), ctypes.byref(ws))
            plan = ("gemm", ws.value)
        elif WGRAD_STRIP and kh == 3 and kw == 3 and lib().zs3_conv_wgrad_strip_plan(
                I(n), I(h), I(w_), I(ho), I(wo), I(kh), I(kw), I(stride), I(pad_h), I(pad_w), I(dil), I(cout), I(cin),
                ctypes.byref(splitk), ctypes.byref(ws)):
            plan = ("strip", ws.value)
        elif WGRAD_PW and kh == 1 and kw == 1 and stride == 1 and pad_h == 0 and pad_w == 0 and (h, w_) == (ho, wo) and \
                lib().zs3_conv_wgrad_pw_plan(I(n * h * w_), I(cout), I(cin), ctypes.byref(splitk), ctypes.byref(ws)):
            plan = ("pw", ws.value)
        else:
            lib().zs3_conv_wgrad_plan(I(n * ho * wo), I(wo), I(cout), I(cin), I(kh * kw), ctypes.byref(splitk), ctypes.byref(ws))
            plan = ("gemm", ws.value)
        _WGRAD_PLAN[key] = plan
    return plan


def consumer_applies_bn(xshape, ldx, wp, stride, pad, dil, prec=None):
    """Can the conv that consumes x (NHWC shape `xshape`, row stride ldx) apply the BatchNorm + ReLU of the layer that produced x
    in its own operand path -- forward AND weight gradient on kernels whose producer waves convert the operand (strip-resident /
    pointwise)?  Then the producing layer skips its BN-apply pass and hands over its raw conv output (functional._ConvBnAct)."""
    prec = PREC_DEFAULT if (prec is None or PREC_DEFAULT == 0) else prec
    n, h, w_, _ = xshape
    ho, wo = conv_out_size(h, wp.kh, stride, pad, dil), conv_out_size(w_, wp.kw, stride, pad, dil)
    cin_valid = min(_round_up(wp.cin, 4), ldx)
    if wp.cin % 4 or ldx % 4 or ACT_DTYPE == BF16:   # (bf16-stored operands are copied, not converted: no transform in the producers)
        return False
    tile = _choose_tile(0, xshape, n * ho * wo, ho, wo, wp.cin_pad, cin_valid, ldx, wp.kh, wp.kw, stride, pad, pad, dil, wp.cout,
                        False, prec, True)
    if tile not in (41, 42, 51, 52) or (tile in (51, 52) and wp.cin % 32):   # (the pointwise kernel transforms whole 32-channel K steps only)
        return False
    return _wgrad_plan(n, h, w_, ho, wo, wp.kh, wp.kw, stride, pad, pad, dil, wp.cout, wp.cin)[0] in ("strip", "pw")


def conv2d_wgrad(dy, x, cout, cin, kh, kw, stride=1, pad_h=0, pad_w=None, dil=1, prec=None, ci_read=None, out=None, x_affine=None):
    """dy: NHWC [N,Ho,Wo,>=cout]; x: NHWC [N,H,W,>=cin] -> dw [cout, kh, kw, cin] (channels_last weight storage).
    x_affine = (scale, shift): x is read through max(x * scale + shift, 0) (strip-resident / pointwise kernels only)."""
    require_gpu(dy, x)
    prec = PREC_DEFAULT if (prec is None or PREC_DEFAULT == 0) else prec
    pad_w = pad_h if pad_w is None else pad_w
    n, ho, wo, _ = dy.shape
    _, h, w_, _ = x.shape
    lddy, ldx = _check_nhwc(dy), _check_nhwc(x)
    co_read = min(_round_up(cout, 4), lddy)
    ci_read = ci_read or min(_round_up(cin, 4), ldx)
    dw = out if out is not None else torch.empty((cout, kh, kw, cin), dtype=torch.float32, device=x.device)
    assert dw.is_contiguous() and dw.numel() == cout * kh * kw * cin
    io = _io(dy) | (_io(x) << 1)
    if io and prec != 1:
        raise ValueError("bf16-stored operands need plain-bf16 products (prec = 1)")
    plan = _wgrad_plan(n, h, w_, ho, wo, kh, kw, stride, pad_h, pad_w, dil, cout, cin, io)
    kind, nws = plan
    xs, xh = x_affine if x_affine is not None else (None, None)
    if x_affine is not None and kind == "gemm":
        raise ValueError("x_affine needs the strip-resident or the pointwise weight-gradient kernel")
    work = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
    if kind == "strip":
        # strip-resident kernel (csrc/conv_wgrad_strip.hip): all nine taps from one LDS-resident strip of x
        check(lib().zs3_conv_wgrad_strip(P(dy), P(x), P(dw), P(work), I(n), I(h), I(w_), I(dil), I(co_read), I(cout),
                                         I(ci_read), I(cin), I(lddy), I(ldx), I(prec), P(zero_page(x.device)), P(xs), P(xh),
                                         I(io), stream()), "zs3_conv_wgrad_strip")
    elif kind == "pw":
        # pointwise kernel (csrc/conv_wgrad_strip.hip): producer waves split both operands once, transposing fragment reads
        check(lib().zs3_conv_wgrad_pw(P(dy), P(x), P(dw), P(work), I(n * h * w_), I(co_read), I(cout), I(ci_read), I(cin),
                                      I(lddy), I(ldx), I(prec), P(zero_page(x.device)), P(xs), P(xh),
                                      I(io), stream()), "zs3_conv_wgrad_pw")
    else:
        check(lib().zs3_conv_wgrad(P(dy), P(x), P(dw), P(work), I(n), I(h), I(w_), I(ho), I(wo), I(kh), I(kw), I(stride),
                                   I(pad_h), I(pad_w), I(dil), I(co_read), I(cout), I(ci_read), I(cin), I(lddy), I(ldx),
                                   I(prec), P(zero_page(x.device)), I(io), stream()), "zs3_conv_wgrad")
    return dw


# ------------------------------------------------------------------------------------------- BN / elementwise
def _rows(t):
    """[..., C] tensor with contiguous channels and uniform row stride -> (M, C, ld)."""
    c = t.shape[-1]
    if t.dim() == 2:
        return t.shape[0], c, (t.stride(0) if t.shape[0] > 1 else max(c, t.stride(0)))
    ld = _check_nhwc(t)
    return t.shape[0] * t.shape[1] * t.shape[2], c, ld


def colstats(x):
    require_gpu(x)
    m, c, ld = _rows(x)
    chunks, rpb = ctypes.c_int(0), ctypes.c_int(0)
    lib().zs3_colstats_plan(I(m), I(c), ctypes.byref(chunks), ctypes.byref(rpb))
    part = torch.empty((chunks.value, 2, c), dtype=torch.float32, device=x.device)
    check(lib().zs3_colstats(P(x), I(ld), I(m), I(c), P(part), I(_io(x)), stream()), "zs3_colstats")
    return part


def _count_args(count):
    """(host count, device count pointer): `count` is a python number or a 1-element fp64 device tensor."""
    if torch.is_tensor(count):
        assert count.dtype == torch.float64 and count.numel() == 1 and count.is_cuda
        return ctypes.c_double(0.0), P(count)
    return ctypes.c_double(count), P(None)


def _partial_args(partial, count):
    """(pointer, chunks, C, host count, device count) of a finalize call: `partial` is the [chunks][2][C] fp32 partial sums, or
    the fp64 SyncBN exchange buffer [2C + 1] (bn_sync_pack + all-reduce: chunks = -1, the count is its last element)."""
    if partial.dtype == torch.float64:
        c = (partial.numel() - 1) // 2
        return partial, -1, c, ctypes.c_double(0.0), P(partial[2 * c:])
    chost, cdev = _count_args(count)
    return partial, partial.shape[0], partial.shape[2], chost, cdev


def bn_sync_pack(partial, count):
    """-> fp64 [sum | second sum | count] (2C + 1) of this rank, the payload of the SyncBN all-reduce."""
    c = partial.shape[2]
    out = torch.empty(2 * c + 1, dtype=torch.float64, device=partial.device)
    check(lib().zs3_bn_sync_pack(P(partial), I(partial.shape[0]), I(c), F(float(count)), P(out), stream()), "zs3_bn_sync_pack")
    return out


_range_flags = {}


def range_flag(device):
    """The sticky range flag of the f16x3 forward on `device` (device int32[1], 0 = fine): raised by zs3_bn_fwd_finalize when
    the batch sums of a layer whose convolution multiplied fp16 hi/lo operands are not finite (an operand beyond +-65504 -- or a
    weight beyond 1023 -- became inf: DESIGN.md section 2), honoured by zs3_sgd_multi (the step is skipped), read and lowered by
    functional.check_forward_range (the forward falls back to bf16x3 products)."""
    key = (device.type, device.index)
    if key not in _range_flags:
        _range_flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _range_flags[key]


def bn_fwd_finalize(partial, count, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked=None,
                    range_flag=None):
    partial, chunks, c, chost, cdev = _partial_args(partial, count)
    out = torch.empty((4, c), dtype=torch.float32, device=partial.device)  # mean, invstd, scale, shift
    base, row = out.data_ptr(), 4 * c           # (row pointers by arithmetic: four tensor views cost 6 us of host time per layer)
    check(lib().zs3_bn_fwd_finalize(P(partial), I(chunks), I(c), chost, cdev, P(gamma), P(beta),
                                    F(eps), F(momentum), P(running_mean), P(running_var), base, base + row,
                                    base + 2 * row, base + 3 * row, P(num_batches_tracked), P(range_flag), stream()),
          "zs3_bn_fwd_finalize")
    return out


def bn_eval_affine(gamma, beta, running_mean, running_var, eps):
    c = running_mean.shape[0]
    out = torch.empty((4, c), dtype=torch.float32, device=running_mean.device)
    check(lib().zs3_bn_eval_affine(P(gamma), P(beta), P(running_mean), P(running_var), F(eps), I(c), P(out[0]), P(out[1]),
                                   P(out[2]), P(out[3]), stream()), "zs3_bn_eval_affine")
    return out


def _drop_args(drop):
    """drop: None or (p, seed) of a dropout fused behind the activation -> the two trailing C arguments"""
    if drop is None:
        return F(0.0), ctypes.c_ulonglong(0)
    return F(drop[0]), ctypes.c_ulonglong(drop[1])


def affine_act(x, scale=None, shift=None, alpha=1.0, res=None, out=None, div=1, act=0, leak=0.2, accumulate=False,
               out_shape=None, mask_out=None, drop=None, out_dtype=None):
    """mask_out: optional uint8 tensor of M*C/4 bytes receiving the sign bits of the pre-activation values.
    drop: (p, seed) -- nn.Dropout fused behind the activation, the mask `dropout(out, p, seed)` would draw."""
    require_gpu(x, scale, shift, res, out)
    m_in, c, ldx = _rows(x)
    if out is None:
        out = torch.empty(out_shape if out_shape is not None else x.shape, dtype=out_dtype or x.dtype, device=x.device)
    m, c2, ldo = _rows(out)
    assert c2 == c and m == m_in * div
    _same_type(x, res)
    ldr = _rows(res)[2] if res is not None else 0
    check(lib().zs3_affine_act(P(x), I(ldx), P(scale), P(shift), F(alpha), P(res), I(ldr), P(out), I(ldo),
                               ctypes.c_long(m), I(c), I(div), I(act), F(leak), I(int(accumulate)), P(mask_out), *_drop_args(drop),
                               I(_io(x, out)), stream()), "zs3_affine_act")
    return out


def bn_bwd_stats(dA, a_out, y, mean, invstd, mask_scale=None, mask_shift=None, mask_bits=None, drop=None):
    m, c, ldd = _rows(dA)
    lda = _rows(a_out)[2] if a_out is not None else 0
    ldy = _rows(y)[2]
    chunks, rpb = ctypes.c_int(0), ctypes.c_int(0)
    lib().zs3_colstats_plan(I(m), I(c), ctypes.byref(chunks), ctypes.byref(rpb))
    part = torch.empty((chunks.value, 2, c), dtype=torch.float32, device=dA.device)
    _same_type(dA, a_out, y)
    check(lib().zs3_bn_bwd_stats(P(dA), I(ldd), P(a_out), I(lda), P(y), I(ldy), P(mean), P(invstd), P(mask_scale), P(mask_shift),
                                 P(mask_bits), I(m), I(c), P(part), *_drop_args(drop), I(_io(dA)), stream()), "zs3_bn_bwd_stats")
    return part


def bn_bwd_finalize(partial, count, use_batch_stats, want_param_grads=True, out=None):
    """-> (dgamma, dbeta, c1, c2).  dgamma / dbeta own their storage so that autograd can adopt them as .grad
    without a copy; out = (dgamma buffer or None, dbeta buffer or None): gradient-bucket slices to write them into."""
    partial, chunks, c, chost, cdev = _partial_args(partial, count)
    dev = partial.device
    if not want_param_grads:      # c1, c2 only (the kernel skips NULL outputs)
        cc = torch.empty((2, c), dtype=torch.float32, device=dev)
        base = cc.data_ptr()
        check(lib().zs3_bn_bwd_finalize(P(partial), I(chunks), I(c), chost, cdev, None, None, base, base + 4 * c,
                                        I(int(use_batch_stats)), stream()), "zs3_bn_bwd_finalize")
        return None, None, cc[0], cc[1]
    og, ob = out if out is not None else (None, None)
    # (a FRESH view of the slice: autograd adopts a gradient as .grad without a copy only when nobody else holds that tensor object,
    # and the registered slice itself stays in functional._grad_buffers)
    dgamma = og.view(c) if og is not None and og.numel() == c and og.dtype == torch.float32 else torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = ob.view(c) if ob is not None and ob.numel() == c and ob.dtype == torch.float32 else torch.empty(c, dtype=torch.float32, device=dev)
    cc = torch.empty((2, c), dtype=torch.float32, device=dev)
    base = cc.data_ptr()
    check(lib().zs3_bn_bwd_finalize(P(partial), I(chunks), I(c), chost, cdev, P(dgamma), P(dbeta),
                                    base, base + 4 * c, I(int(use_batch_stats)), stream()), "zs3_bn_bwd_finalize")
    return dgamma, dbeta, cc[0], cc[1]


def bn_act_bwd(dA, a_out, y, mean, invstd, gamma, c1, c2, dy=None, dres=None, dres_accumulate=False, act=1, leak=0.2,
               want_dy=True, mask_scale=None, mask_shift=None, mask_bits=None, drop=None):
    require_gpu(dA, a_out, y, dy, dres)
    m, c, ldd = _rows(dA)
    if want_dy and dy is None:
        dy = torch.empty(dA.shape, dtype=dA.dtype, device=dA.device)
    _same_type(dA, a_out, y, dy, dres)
    lda = _rows(a_out)[2] if a_out is not None else 0
    ldy = _rows(y)[2] if y is not None else 0
    ldo = _rows(dy)[2] if dy is not None else 0
    ldr = _rows(dres)[2] if dres is not None else 0
    check(lib().zs3_bn_act_bwd(P(dA), I(ldd), P(a_out), I(lda), P(y), I(ldy), P(mean), P(invstd), P(gamma), P(c1), P(c2),
                               P(mask_scale), P(mask_shift), P(mask_bits), P(dy), I(ldo), P(dres), I(ldr), I(int(dres_accumulate)), ctypes.c_long(m), I(c), I(act),
                               F(leak), *_drop_args(drop), I(3 if dA.dtype == BF16 else 0), stream()), "zs3_bn_act_bwd")
    return dy


def group_colsum(x, groups, scale=1.0, out=None):
    m, c, ld = _rows(x)
    r = m // groups
    if out is None:
        out = torch.empty((groups, c), dtype=x.dtype, device=x.device)
    _same_type(x, out)
    check(lib().zs3_group_colsum(P(x), I(ld), I(groups), I(r), I(c), F(scale), P(out), I(_rows(out)[2]), I(3 if x.dtype == BF16 else 0),
                                 stream()), "zs3_group_colsum")
    return out


# ------------------------------------------------------------------------------------------- pool / resize
def maxpool_fwd(x, k=3, stride=2, pad=1):
    n, h, w_, c = x.shape
    ho, wo = conv_out_size(h, k, stride, pad, 1), conv_out_size(w_, k, stride, pad, 1)
    out = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
    check(lib().zs3_maxpool_fwd(P(x), I(_check_nhwc(x)), P(out), I(c), P(idx), I(n), I(h), I(w_), I(ho), I(wo), I(c), I(k),
                                I(stride), I(pad), I(3 if x.dtype == BF16 else 0), stream()), "zs3_maxpool_fwd")
    return out, idx


def maxpool_bwd(dy, idx, in_hw, k=3, stride=2, pad=1):
    n, ho, wo, c = dy.shape
    h, w_ = in_hw
    dx = torch.empty((n, h, w_, c), dtype=dy.dtype, device=dy.device)
    check(lib().zs3_maxpool_bwd(P(dy), I(_check_nhwc(dy)), P(idx), P(dx), I(c), I(n), I(h), I(w_), I(ho), I(wo), I(c),
                                I(k), I(stride), I(pad), I(3 if dy.dtype == BF16 else 0), stream()), "zs3_maxpool_bwd")
    return dx


def bilinear_fwd(x, size, out=None):
    n, h, w_, c = x.shape
    ho, wo = size
    if out is None:
        out = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    _same_type(x, out)
    check(lib().zs3_bilinear_fwd(P(x), I(_check_nhwc(x)), P(out), I(_check_nhwc(out)), I(n), I(h), I(w_), I(ho), I(wo),
                                 I(c), I(3 if x.dtype == BF16 else 0), stream()), "zs3_bilinear_fwd")
    return out


def bilinear_bwd(dout, in_hw, out=None, accumulate=False):
    n, ho, wo, c = dout.shape
    h, w_ = in_hw
    if out is None:
        out = torch.empty((n, h, w_, c), dtype=dout.dtype, device=dout.device)
    _same_type(dout, out)
    check(lib().zs3_bilinear_bwd(P(dout), I(_check_nhwc(dout)), P(out), I(_check_nhwc(out)), I(n), I(h), I(w_), I(ho),
                                 I(wo), I(c), I(int(accumulate)), I(3 if dout.dtype == BF16 else 0), stream()), "zs3_bilinear_bwd")
    return out


# ------------------------------------------------------------------------------------------- misc
def dropout(x, p, seed, out=None, row_idx=None, seed_dev=None):
    m, c, ld = _rows(x)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _same_type(x, out)
    check(lib().zs3_dropout(P(x), I(ld), P(out), I(_rows(out)[2]), ctypes.c_long(m), I(c), F(p),
                            ctypes.c_ulonglong(seed), P(row_idx), P(seed_dev), I(3 if x.dtype == BF16 else 0), stream()), "zs3_dropout")
    return out


def uniform(shape, seed, device, out=None, seed_dev=None):
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=device)
    check(lib().zs3_uniform(P(out), ctypes.c_long(out.numel()), ctypes.c_ulonglong(seed), P(seed_dev), stream()), "zs3_uniform")
    return out


def counter_add(counter, v=1):
    check(lib().zs3_counter_add(P(counter), ctypes.c_long(v), stream()), "zs3_counter_add")


def counter_add2(c0, v0, c1, v1):
    check(lib().zs3_counter_add2(P(c0), ctypes.c_long(v0), P(c1), ctypes.c_long(v1), stream()), "zs3_counter_add2")


def gather_cat_noise(a, idx, ca, cb, ldo, n, seed, seed_dev=None, noise_key=None):
    _fp32_rows(a)
    """[a[idx] | U[0,1)^cb | 0] rows: zs3_gather_cat on a zs3_uniform tensor without materialising the noise.  noise_key:
    optional int64 [n] -- the noise of row r is the noise of "pixel" noise_key[r] (duplicate samples share it)."""
    out = torch.empty((n, ldo), dtype=torch.float32, device=a.device)
    check(lib().zs3_gather_cat_noise(P(a), I(a.stride(0)), P(idx), I(ca), I(cb), P(out), I(ldo), ctypes.c_long(n),
                                     ctypes.c_ulonglong(seed), P(seed_dev), P(noise_key), stream()), "zs3_gather_cat_noise")
    return out


def dropout_act_bwd(dy, h, p, seed, leak, row_idx=None, seed_dev=None):
    _fp32_rows(dy, h)
    m, c, ldd = _rows(dy)
    out = torch.empty((m, c), dtype=torch.float32, device=dy.device)
    check(lib().zs3_dropout_act_bwd(P(dy), I(ldd), P(h), I(_rows(h)[2]), P(out), I(c), ctypes.c_long(m), I(c), F(p),
                                    ctypes.c_ulonglong(seed), P(row_idx), P(seed_dev), F(leak), stream()),
          "zs3_dropout_act_bwd")
    return out


def zeros(shape, dtype, device):
    """torch.zeros through the library (zs3_fill_zero): a recorded plan replays the fill"""
    t = torch.empty(shape, dtype=dtype, device=device)
    check(lib().zs3_fill_zero(P(t), ctypes.c_long(t.numel() * t.element_size()), stream()), "zs3_fill_zero")
    return t


def pad_rows(t, cp):
    """[..., C] rows (contiguous channels, uniform row stride) -> a dense [..., cp] tensor's [..., :C] view whose pad channels are zero"""
    m, c, ld = _rows(t)
    buf = torch.empty(t.shape[:-1] + (cp,), dtype=t.dtype, device=t.device)
    check(lib().zs3_pad_rows(P(t), I(ld), I(c), P(buf), I(cp), ctypes.c_long(m), I(3 if t.dtype == BF16 else 0), stream()), "zs3_pad_rows")
    return buf[..., :c]


def colsum(x, out=None):
    _fp32_rows(x, out)
    m, c, ld = _rows(x)
    if out is None:
        out = torch.empty(c, dtype=torch.float32, device=x.device)
    check(lib().zs3_colsum(P(x), I(ld), I(m), I(c), P(out), stream()), "zs3_colsum")
    return out


def nearest_rows(src_chw, size, ld=None):
    c, h, w_ = src_chw.shape
    ho, wo = size
    assert src_chw.is_contiguous()
    ld = ld or c
    rows = torch.zeros((ho * wo, ld), dtype=torch.float32, device=src_chw.device) if ld != c else torch.empty(
        (ho * wo, ld), dtype=torch.float32, device=src_chw.device)
    check(lib().zs3_nearest_rows(P(src_chw), I(c), I(h), I(w_), I(ho), I(wo), P(rows), I(ld), stream()), "zs3_nearest_rows")
    return rows


def label_order(target, size):
    """target: [B,H,W] float32 / int64 label maps -> (tgt_l [B,npix], tgt_cls [B,npix], hist [B,256], order [B,npix]), int64:
    nearest resize to `size`, 255 -> class 0, class histogram, stable argsort by class -- one launch (zs3_label_order)."""
    require_gpu(target)
    if target.dtype not in (torch.float32, torch.int64):
        target = target.float()
    target = target.contiguous()
    b, h, w_ = target.shape
    ho, wo = size
    out = torch.empty((3, b, ho * wo), dtype=torch.int64, device=target.device)
    hist = torch.empty((b, 256), dtype=torch.int64, device=target.device)
    check(lib().zs3_label_order(P(target), I(int(target.dtype == torch.int64)), I(b), I(h), I(w_), I(ho), I(wo), P(out[0]),
                                P(out[1]), P(hist), P(out[2]), stream()), "zs3_label_order")
    return out[0], out[1], hist, out[2]


def _fp32_rows(*tensors):
    """the row kernels (gather / scatter / index_add / gather_cat) move fp32 rows and nothing else: a bf16 tensor handed to them would be
    read as fp32, twice past its end"""
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise TypeError(f"row kernels take fp32 rows, got {t.dtype}")


def gather_cat(a, idx, ca, b, cb, ldo):
    _fp32_rows(a, b)
    n = b.shape[0]
    out = torch.empty((n, ldo), dtype=torch.float32, device=a.device)
    check(lib().zs3_gather_cat(P(a), I(a.stride(0)), P(idx), I(ca), P(b), I(b.stride(0)), I(cb), P(out), I(ldo),
                               ctypes.c_long(n), stream()), "zs3_gather_cat")
    return out


def gather_rows(src, idx, c=None):
    _fp32_rows(src)
    c = c or src.shape[1]
    n = idx.shape[0]
    out = torch.empty((n, c), dtype=torch.float32, device=src.device)
    check(lib().zs3_gather_rows(P(src), I(src.stride(0)), P(idx), P(out), I(c), ctypes.c_long(n), I(c), stream()),
          "zs3_gather_rows")
    return out


def scatter_rows(src, idx, out, c=None):
    _fp32_rows(src, out)
    c = c or src.shape[1]
    check(lib().zs3_scatter_rows(P(src), I(src.stride(0)), P(idx), P(out), I(out.stride(0)), ctypes.c_long(idx.shape[0]),
                                 I(c), stream()), "zs3_scatter_rows")
    return out


def index_add_rows(src, idx, out, c=None):
    _fp32_rows(src, out)
    c = c or src.shape[1]
    check(lib().zs3_index_add_rows(P(src), I(src.stride(0)), P(idx), P(out), I(out.stride(0)), I(idx.shape[0]), I(c),
                                   stream()), "zs3_index_add_rows")
    return out
