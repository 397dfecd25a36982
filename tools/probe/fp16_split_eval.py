"""VERDICT r3 #3c: would an fp16 hi/lo operand split (11 + 11 mantissa bits) make the three-MFMA product "fp32" where the bf16
split (8 + 8) is 2^-16-class?  CPU emulation of ONE layer's products (the MFMA accumulates in fp32, emulated by an fp32 matmul of
the split operands); relative L2 error against the fp64 product of the fp32 operands.  Operand magnitudes cover what the step holds:
post-ReLU activations O(1), weights O(1e-2), back-propagated gradients O(1e-3 ... 1e-8) (the loss is divided by B x 513^2 pixels).
usage: python tools/probe/fp16_split_eval.py"""
import torch

torch.manual_seed(0)
M, K, N = 2048, 1024, 256


def split(x, dt, scale=1.0):
    xs = x * scale
    hi = xs.to(dt)
    lo = (xs - hi.float()).to(dt)
    return hi.float(), lo.float()


def prod3(a, b, dt, sa=1.0, sb=1.0):
    ah, al = split(a, dt, sa)
    bh, bl = split(b, dt, sb)
    acc = (al @ bh.t()) + (ah @ bl.t()) + (ah @ bh.t())      # fp32 accumulate, the kernel's product order
    return acc / (sa * sb)


def rel(y, ref):
    return float((y.double() - ref).norm() / ref.norm())


def pow2_scale(x, target_exp=8):
    """power of two that brings max|x| to 2^target_exp (fp16's largest finite value is 2^16 - 32)"""
    e = torch.floor(torch.log2(x.abs().max())).item()
    return 2.0 ** (target_exp - e)


w = torch.randn(N, K) * (2.0 / K) ** 0.5
print("| A operand | bf16 hi/lo x3 | fp16 hi/lo x3, unscaled | fp16 hi/lo x3, power-of-two scale per tensor | plain fp32 matmul |")
print("|---|---|---|---|---|")
for name, a in (("activations, post-ReLU, O(1)", torch.relu(torch.randn(M, K))),
                ("gradients, O(1e-3)", torch.randn(M, K) * 1e-3),
                ("gradients, O(1e-6)", torch.randn(M, K) * 1e-6),
                ("gradients, O(1e-8) with 1e3 dynamic range across rows", torch.randn(M, K) * 1e-8 * torch.logspace(0, 3, M)[:, None]),
                ("gradients, O(1e-6) with 1e6 dynamic range across rows", torch.randn(M, K) * 1e-9 * torch.logspace(0, 6, M)[:, None])):
    ref = a.double() @ w.double().t()
    row = [rel(prod3(a, w, torch.bfloat16), ref), rel(prod3(a, w, torch.float16), ref),
           rel(prod3(a, w, torch.float16, pow2_scale(a), pow2_scale(w)), ref), rel(a @ w.t(), ref)]
    # per-row error of the scaled fp16 form: rows far below the tensor's maximum lose their lo part to fp16's subnormal floor
    y = prod3(a, w, torch.float16, pow2_scale(a), pow2_scale(w))
    rows = ((y.double() - ref).norm(dim=1) / ref.norm(dim=1))
    print(f"| {name} | " + " | ".join(f"{v:.1e}" for v in row) + f" | (scaled fp16: worst row {rows.max():.1e}, median row {rows.median():.1e})")
