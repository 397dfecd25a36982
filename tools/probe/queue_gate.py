"""What gates a chain of small launches beside a stream of convolutions?  Stream B: 300 dependent launches of the generator's second
GEMM (zs3_gmmn_mlp_fwd2: 64 workgroups x 256 threads, 50 KB of LDS, ~7 us each) -- the shape of the GMMN step's update chain.
Stream A: one convolution kernel of the frozen feature pass, repeated back to back.  Prints B's time per launch alone and beside each
kind of A, and A's time per launch beside B.   python tools/probe/queue_gate.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from zs3_amd import functional as Fz
from zs3_amd import ops
from zs3_amd._lib import I, P, check, lib

dev = torch.device("cuda:0")
Fz.warm_streams(dev)
sa, sb = Fz.feature_stream(dev), torch.cuda.current_stream(dev)
g = torch.Generator(device=dev).manual_seed(0)
hd = torch.randn(128, 256, device=dev, generator=g)
w2 = torch.randn(256, 256, device=dev, generator=g) * 0.05
wp2 = ops.prep_weight(w2)
bias = torch.zeros(256, device=dev)
gen = torch.empty(128, 256, device=dev)


ROWS = int(os.environ.get("CHAIN_ROWS", "128"))      # 128 rows = 64 workgroups per launch, 64 = 32, 32 = 16


def chain(n=300):
    for _ in range(n):
        check(lib().zs3_gmmn_mlp_fwd2(P(hd), I(256), P(wp2.f_pk), I(wp2.cin_pad // 32), P(bias), P(gen), I(256), I(ROWS), I(256), I(256),
                                      None, I(0), None, None, sb.cuda_stream), "fwd2")


x33 = torch.randn(16, 33, 33, 1024, device=dev, generator=g)
x33b = torch.randn(16, 33, 33, 256, device=dev, generator=g)
x129 = torch.randn(16, 129, 129, 256, device=dev, generator=g)
w_1024_256 = ops.prep_weight(torch.randn(256, 1024, 1, 1, device=dev, generator=g) * 0.02, f16_forward=True)
w_256_1024 = ops.prep_weight(torch.randn(1024, 256, 1, 1, device=dev, generator=g) * 0.02, f16_forward=True)
w_3x3 = ops.prep_weight(torch.randn(256, 256, 3, 3, device=dev, generator=g) * 0.02, f16_forward=True)
sc, sh = torch.rand(1024, device=dev) + 0.5, torch.randn(1024, device=dev) * 0.1
kinds = {
    "conv_igemm_dma 1024->256 @33^2 (138 tiles)": lambda: ops.conv2d_fwd(x33, w_1024_256, 1, 0, 1, tile_cfg=31, want_stats=True, prec=4),
    "conv_pw 256->1024 @33^2 (persistent, 2 per CU)": lambda: ops.conv2d_fwd(x33b, w_256_1024, 1, 0, 1, tile_cfg=52, want_stats=True, prec=4),
    "conv_halo 3x3 256->256 @33^2 (182 tiles)": lambda: ops.conv2d_fwd(x33b, w_3x3, 1, 1, 1, tile_cfg=42, want_stats=True, prec=4),
    "conv_halo 3x3 256->256 @129^2 (2774 tiles)": lambda: ops.conv2d_fwd(x129, w_3x3, 1, 1, 1, tile_cfg=42, want_stats=True, prec=4),
    "conv_igemm 64x64 1024->256 @33^2 (1092 tiles)": lambda: ops.conv2d_fwd(x33, w_1024_256, 1, 0, 1, tile_cfg=14, want_stats=True, prec=4),
    "affine_act 33^2 x 1024": lambda: ops.affine_act(x33, sc, sh, act=1),
}


def timed_chain(n=300):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(sb)
    chain(n)
    e1.record(sb)
    return e0, e1


chain(50)
torch.cuda.synchronize()
e0, e1 = timed_chain()
torch.cuda.synchronize()
alone = e0.elapsed_time(e1) / 300 * 1e3
print(f"chain alone: {alone:6.1f} us per launch")
for pw_wgs in (256, 192):
    lib().zs3_conv_pw_set_wgs(pw_wgs)
    for name, fn in kinds.items():
        if pw_wgs == 192 and "conv_pw" not in name:
            continue
        with torch.cuda.stream(sa):
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 0
        with torch.cuda.stream(sa):
            a0.record(sa)
            for _ in range(400):
                fn()
                reps += 1
            a1.record(sa)
        e0, e1 = timed_chain()
        torch.cuda.synchronize()
        beside = e0.elapsed_time(e1) / 300 * 1e3
        print(f"beside {name:52s} (pw wgs {pw_wgs}): chain {beside:7.1f} us per launch ({beside / alone:4.1f}x), "
              f"A {a0.elapsed_time(a1) / reps * 1e3:7.1f} us per launch", flush=True)
lib().zs3_conv_pw_set_wgs(256)
# the strip-resident kernel on a capped grid (zs3_conv_halo_set_wgs): same results, and what the chain does beside it
name = "conv_halo 3x3 256->256 @129^2 (2774 tiles)"
ref = kinds[name]()
torch.cuda.synchronize()
for wgs in (0, 256, 224, 192):
    lib().zs3_conv_halo_set_wgs(wgs)
    out = kinds[name]()
    torch.cuda.synchronize()
    same = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        a0.record(sa)
        for _ in range(60):
            kinds[name]()
        a1.record(sa)
    e0, e1 = timed_chain()
    torch.cuda.synchronize()
    beside = e0.elapsed_time(e1) / 300 * 1e3
    print(f"beside {name} on {wgs or 'one per tile'} workgroups: identical {same}; chain {beside:7.1f} us per launch "
          f"({beside / alone:4.1f}x), A {a0.elapsed_time(a1) / 60 * 1e3:7.1f} us per launch", flush=True)
lib().zs3_conv_halo_set_wgs(0)
