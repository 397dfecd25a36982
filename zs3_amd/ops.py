"""Thin, non-differentiable wrappers over the C ABI (one Python function per kernel family).

Internal activation layout is NHWC: a tensor [N, H, W, C] whose last dim is contiguous and whose
pixel stride (`ld`) may exceed C (channel-slice views of a wider buffer).  Logical-NCHW tensors with
channels_last strides are the same memory: `nhwc(t)` / `nchw(t)` convert without copying.
"""
import ctypes
from dataclasses import dataclass

import torch

from ._lib import F, I, P, check, lib, require_gpu, stream

PREC_DEFAULT = 3  # bf16x3 split; 1 = plain bf16 inputs


def _round_up(a, b):
    return (a + b - 1) // b * b


def nhwc(t):
    """Logical NCHW (channels_last memory) -> [N,H,W,C] view; copies only if the memory is not NHWC."""
    v = t.permute(0, 2, 3, 1)
    if v.stride(3) != 1 or v.stride(2) % 4 != 0 and v.shape[3] > 1:
        v = v.contiguous()
    return v


def nchw(t):
    return t.permute(0, 3, 1, 2)


def _check_nhwc(t):
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
    f_hi: torch.Tensor
    f_lo: torch.Tensor
    t_hi: torch.Tensor
    t_lo: torch.Tensor
    cout: int
    cin: int
    kh: int
    kw: int
    cin_pad: int
    cout_pad: int


def prep_weight(w, need_t=True, cin_pad=None):
    """w: [Cout, Cin, KH, KW] parameter (any strides) or [Cout, Cin] linear weight -> bf16 hi/lo planes."""
    require_gpu(w)
    if w.dim() == 2:
        w = w[:, :, None, None]
    cout, cin, kh, kw = w.shape
    wl = w.detach().permute(0, 2, 3, 1).contiguous()  # [Cout][KH][KW][Cin]; a no-op view for channels_last params
    cin_pad = cin_pad or _round_up(cin, 32)
    cout_pad = _round_up(cout, 32)
    taps = kh * kw
    dev = w.device
    f_hi = torch.empty((cout, taps * cin_pad), dtype=torch.bfloat16, device=dev)
    f_lo = torch.empty_like(f_hi)
    if need_t:
        t_hi = torch.empty((cin, taps * cout_pad), dtype=torch.bfloat16, device=dev)
        t_lo = torch.empty_like(t_hi)
    else:
        t_hi = t_lo = None
    check(lib().zs3_prep_weight(P(wl), P(f_hi), P(f_lo), P(t_hi), P(t_lo), I(cout), I(taps), I(cin), I(cin_pad),
                                I(cout_pad), stream()), "zs3_prep_weight")
    return WeightPlanes(f_hi, f_lo, t_hi, t_lo, cout, cin, kh, kw, cin_pad, cout_pad)


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv_igemm(x, w_hi, w_lo, *, ho, wo, cin_pad, cin_valid, kh, kw, stride, pad_h, pad_w, dil, ncols, out=None,
               scale=None, shift=None, res=None, want_stats=False, act=0, leak=0.2, accumulate=False, dgrad=False,
               prec=None, tile_cfg=0):
    """Raw launcher.  x: NHWC [N,H,W,*]; returns (y [N,ho,wo,ncols] or `out`, stat_partial or None)."""
    require_gpu(x, w_hi, out, scale, shift, res)
    prec = prec or PREC_DEFAULT
    n, h, w_, _ = x.shape
    ldx = _check_nhwc(x)
    if out is None:
        out = torch.empty((n, ho, wo, ncols), dtype=torch.float32, device=x.device)
    ldy = _check_nhwc(out)
    ldr = _check_nhwc(res) if res is not None else 0
    m = n * ho * wo
    stat = None
    if want_stats:
        mt = lib().zs3_conv_igemm_mtiles(I(m), I(ncols), I(tile_cfg))
        stat = torch.empty((mt, 2, ncols), dtype=torch.float32, device=x.device)
    check(lib().zs3_conv_igemm(P(x), P(w_hi), P(w_lo), P(out), P(scale), P(shift), P(res), P(stat), I(n), I(h), I(w_),
                               I(ho), I(wo), I(cin_pad), I(cin_valid), I(ldx), I(kh), I(kw), I(stride), I(pad_h),
                               I(pad_w), I(dil), I(ncols), I(ldy), I(ldr), I(act), F(leak), I(int(accumulate)),
                               I(int(dgrad)), I(prec), I(tile_cfg), stream()), "zs3_conv_igemm")
    return out, stat


def conv2d_fwd(x, wp, stride=1, pad=0, dil=1, **kw):
    """x: NHWC [N,H,W,C>=wp.cin] (pad channels, if any, must be zero)."""
    n, h, w_, c = x.shape
    ho = conv_out_size(h, wp.kh, stride, pad, dil)
    wo = conv_out_size(w_, wp.kw, stride, pad, dil)
    cin_valid = min(_round_up(wp.cin, 8), _check_nhwc(x))
    return conv_igemm(x, wp.f_hi, wp.f_lo, ho=ho, wo=wo, cin_pad=wp.cin_pad, cin_valid=cin_valid, kh=wp.kh, kw=wp.kw,
                      stride=stride, pad_h=pad, pad_w=pad, dil=dil, ncols=wp.cout, **kw)


def conv2d_dgrad(dy, wp, in_hw, stride=1, pad=0, dil=1, **kw):
    """dy: NHWC [N,Ho,Wo,C>=wp.cout] (channels beyond cout zero) -> dx [N,H,W,wp.cin]."""
    h, w_ = in_hw
    cin_valid = min(_round_up(wp.cout, 8), _check_nhwc(dy))
    out, _ = conv_igemm(dy, wp.t_hi, wp.t_lo, ho=h, wo=w_, cin_pad=wp.cout_pad, cin_valid=cin_valid, kh=wp.kh,
                        kw=wp.kw, stride=stride, pad_h=pad, pad_w=pad, dil=dil, ncols=wp.cin, dgrad=True, **kw)
    return out
