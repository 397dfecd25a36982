"""Numpy model of csrc/conv_halo.hip's data movement: the same address formulas (strip rows, tap windows, per-lane masks,
XOR swizzles, weight-chunk mapping, MFMA fragment ownership) executed lane by lane on the CPU and compared with a direct
convolution.  It checks the index algebra of the kernel design, not the compiled kernel (tests/test_gpu_ops.py does that).
usage: python tools/probe/halo_model.py"""
import itertools
import numpy as np

BSLOT, NSLOT = 8192, 2
OFF_ZERO = NSLOT * BSLOT
OFF_TAPS = OFF_ZERO + 128
OFF_STRIP = OFF_ZERO + 256


def bf16_round(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


def to_bits(x):   # fp32 that is exactly a bf16 -> uint16
    return (x.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)


def from_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def geometry(N, H, W, cin_valid, cin_pad, KH, KW, pad, dil, dgrad, prec, bm):
    T = KH * KW
    sgn = -1 if dgrad else 1
    offs = [sgn * ((th * dil - pad) * W + (tw * dil - pad)) for th in range(KH) for tw in range(KW)]
    omin, omax = min(0, min(offs)), max(0, max(offs))
    S = bm + omax - omin
    s_pad = (S + 63) // 64 * 64
    npass = s_pad // 64
    npg = (npass + 5) // 6
    ch = 16 if prec == 3 else 32
    nch = (cin_valid + ch - 1) // ch
    assert nch * ch <= cin_pad and npg <= 3 and T == 9
    return dict(T=T, sgn=sgn, off_min=omin, s_pad=s_pad, npass=npass, npg=npg, nch=nch, ns=(nch * T + 1) & ~1, ch=ch, offs=offs)


def pack_weight(w, cin_pad):
    """[cout][taps][cin] fp32 -> uint16 [cout][K/32][2][32] (zs3_prep_weight)"""
    cout, taps, cin = w.shape
    K = taps * cin_pad
    wp = np.zeros((cout, taps, cin_pad), np.float32)
    wp[:, :, :cin] = w
    wp = wp.reshape(cout, K)
    hi = bf16_round(wp)
    lo = bf16_round(wp - hi)
    out = np.zeros((cout, K // 32, 2, 32), np.uint16)
    out[:, :, 0, :] = to_bits(hi).reshape(cout, K // 32, 32)
    out[:, :, 1, :] = to_bits(lo).reshape(cout, K // 32, 32)
    return out


def run_tile(x, wpk, N, H, W, cin_valid, cin_pad, ldx, KH, KW, pad, dil, dgrad, prec, bm, ncols, mt, nt):
    g = geometry(N, H, W, cin_valid, cin_pad, KH, KW, pad, dil, dgrad, prec, bm)
    T, CH = g["T"], g["ch"]
    M = N * H * W
    m0, n0 = mt * bm, nt * 128
    strip_bytes = g["s_pad"] * 64
    lds = np.zeros(OFF_STRIP + 2 * strip_bytes, np.uint8)
    lds16 = lds.view(np.uint16)
    wbytes = wpk.reshape(wpk.shape[0], -1).view(np.uint8)     # [cout][K*4 bytes]
    TM = bm // 64
    acc = np.zeros((bm, 128), np.float64)
    xf = x.reshape(M, ldx)

    # ---- producers
    def strip_fill(c, sb):
        for p_ in range(g["npass"]):
            for pl in range(256):
                prow, cq = pl >> 2, pl & 3
                q = m0 + g["off_min"] + p_ * 64 + prow
                q = min(max(q, 0), M - 1)
                ch0 = c * CH + cq * (CH // 4)
                s = p_ * 64 + prow
                sw = (s >> 2) & 3
                row = OFF_STRIP + sb * strip_bytes + s * 64
                if prec == 3:
                    v = xf[q, ch0:ch0 + 4] if ch0 < cin_valid else np.zeros(4, np.float32)
                    hi = bf16_round(v)
                    lo = bf16_round(v - hi)
                    o = (((cq >> 1) ^ sw) << 4) + (cq & 1) * 8
                    lds16[(row + o) // 2:(row + o) // 2 + 4] = to_bits(hi)
                    lds16[(row + (o ^ 32)) // 2:(row + (o ^ 32)) // 2 + 4] = to_bits(lo)
                else:
                    v = np.zeros(8, np.float32)
                    for k in range(2):
                        if ch0 + 4 * k < cin_valid:
                            v[4 * k:4 * k + 4] = xf[q, ch0 + 4 * k:ch0 + 4 * k + 4]
                    o = (cq ^ sw) << 4
                    lds16[(row + o) // 2:(row + o) // 2 + 8] = to_bits(bf16_round(v))

    def weight_fill(c, t, slot):
        kofs = t * cin_pad + c * CH
        uoff = (kofs >> 5) * 128 + ((((kofs >> 4) & 1) * 32) if prec == 3 else 0)
        for pl in range(256):
            prow, cq = pl >> 2, pl & 3
            qoff = ((cq & 1) * 16 + (cq >> 1) * 64) if prec == 3 else cq * 16
            for e in range(2):
                wr = prow + 64 * e
                col = n0 + wr
                dst = slot * BSLOT + wr * 64 + ((cq ^ ((wr >> 2) & 3)) << 4)
                if col < ncols and c < g["nch"]:
                    lds[dst:dst + 16] = wbytes[col, uoff + qoff:uoff + qoff + 16]
                else:
                    lds[dst:dst + 16] = 0

    taps = np.array(g["offs"], np.int64)
    # ---- consumer masks
    def mask_of(m):
        if m >= M:
            return 0
        n, rem = divmod(m, H * W)
        y, xx = divmod(rem, W)
        mk = 0
        for t in range(T):
            th, tw = divmod(t, KW)
            iy, ix = y + g["sgn"] * (th * dil - pad), xx + g["sgn"] * (tw * dil - pad)
            if 0 <= iy < H and 0 <= ix < W:
                mk |= 1 << t
        return mk

    def frag(addr):
        return from_bits(lds16[addr // 2:addr // 2 + 8]).astype(np.float64)

    strip_fill(0, 0)
    for s in range(g["nch"] * T):
        c, t = divmod(s, T)
        if t == 0 and c + 1 < g["nch"]:
            strip_fill(c + 1, (c + 1) & 1)     # (the kernel spreads this over the chunk's intervals)
        weight_fill(c, t, s % NSLOT)
        for wave in range(4):
            wm, wn = wave >> 1, wave & 1
            A = np.zeros((TM, 32, 16, 2))      # [row block][row][k][hi/lo]
            B = np.zeros((2, 32, 16, 2))
            for lane in range(64):
                lr, kh = lane & 31, lane >> 5
                rb = wm * (bm // 2) + lr - g["off_min"]
                s0 = rb + int(taps[t])
                a0 = OFF_STRIP + (c & 1) * strip_bytes + s0 * 64 + ((kh ^ ((s0 >> 2) & 3)) << 4)
                for i in range(TM):
                    m = m0 + wm * (bm // 2) + i * 32 + lr
                    addr = a0 + i * 2048 if (mask_of(m) >> t) & 1 else OFF_ZERO
                    A[i, lr, 8 * kh:8 * kh + 8, 0] = frag(addr)
                    A[i, lr, 8 * kh:8 * kh + 8, 1] = frag(addr ^ 32)
                boff = (wn * 64 + lr) * 64 + ((kh ^ ((lr >> 2) & 3)) << 4)
                for j in range(2):
                    addr = (s % NSLOT) * BSLOT + boff + j * 2048
                    B[j, lr, 8 * kh:8 * kh + 8, 0] = frag(addr)
                    B[j, lr, 8 * kh:8 * kh + 8, 1] = frag(addr ^ 32)
            for i in range(TM):
                for j in range(2):
                    if prec == 3:
                        d = (A[i, :, :, 1] @ B[j, :, :, 0].T + A[i, :, :, 0] @ B[j, :, :, 1].T + A[i, :, :, 0] @ B[j, :, :, 0].T)
                    else:   # two K16 halves: "hi" slot = channels 0..15, "lo" slot = channels 16..31
                        d = A[i, :, :, 0] @ B[j, :, :, 0].T + A[i, :, :, 1] @ B[j, :, :, 1].T
                    r0 = wm * (bm // 2) + i * 32
                    acc[r0:r0 + 32, wn * 64 + j * 32:wn * 64 + j * 32 + 32] += d
    return acc


def reference(x, w, N, H, W, cin, KH, KW, pad, dil, dgrad):
    """x [N,H,W,cin]; w [cout][taps][cin] -> y [M][cout] (same-size conv; dgrad: mirrored taps with the given operand)"""
    cout = w.shape[0]
    y = np.zeros((N, H, W, cout))
    sgn = -1 if dgrad else 1
    for th, tw in itertools.product(range(KH), range(KW)):
        dy, dx = sgn * (th * dil - pad), sgn * (tw * dil - pad)
        for yy in range(H):
            iy = yy + dy
            if not 0 <= iy < H:
                continue
            for xx in range(W):
                ix = xx + dx
                if 0 <= ix < W:
                    y[:, yy, xx, :] += x[:, iy, ix, :cin].astype(np.float64) @ w[:, th * KW + tw, :].astype(np.float64).T
    return y.reshape(N * H * W, cout)


def check(N, H, W, cin, cout, KH, KW, dil, dgrad, prec, bm, ldx=None, seed=0):
    rng = np.random.default_rng(seed)
    pad = dil * (KH // 2)
    cin_valid = (cin + 3) // 4 * 4
    cin_pad = (cin + 31) // 32 * 32
    ldx = ldx or cin_valid
    x = np.zeros((N, H, W, ldx), np.float32)
    x[..., :cin] = rng.standard_normal((N, H, W, cin), dtype=np.float32)
    w = (rng.standard_normal((cout, KH * KW, cin), dtype=np.float32) / np.sqrt(cin * KH * KW)).astype(np.float32)
    wpk = pack_weight(w, cin_pad)
    ref = reference(x, w, N, H, W, cin, KH, KW, pad, dil, dgrad)
    M = N * H * W
    worst = 0.0
    for mt in range((M + bm - 1) // bm):
        for nt in range((cout + 127) // 128):
            acc = run_tile(x, wpk, N, H, W, cin_valid, cin_pad, ldx, KH, KW, pad, dil, dgrad, prec, bm, cout, mt, nt)
            rows = min(bm, M - mt * bm)
            cols = min(128, cout - nt * 128)
            r = ref[mt * bm:mt * bm + rows, nt * 128:nt * 128 + cols]
            err = np.abs(acc[:rows, :cols] - r).max() / np.abs(ref).max()
            worst = max(worst, err)
            assert np.abs(acc[rows:, :]).max(initial=0.0) == 0.0, "rows past M must stay zero"
    tol = 2e-5 if prec == 3 else 3e-2
    print(f"N{N} {H}x{W} {cin}->{cout} k{KH}x{KW} d{dil} dgrad={dgrad} prec={prec} bm={bm}: rel err {worst:.2e}")
    assert worst < tol


if __name__ == "__main__":
    check(1, 7, 9, 32, 128, 3, 3, 1, False, 3, 256)
    check(2, 11, 13, 20, 40, 3, 3, 2, False, 3, 192)          # ragged channels, two row tiles, column tail
    check(1, 9, 8, 48, 128, 3, 3, 1, True, 3, 256)            # dgrad: mirrored taps
    check(1, 13, 11, 64, 130, 3, 3, 4, False, 1, 192)         # plain bf16, 32-channel chunks, two column tiles
    check(1, 12, 12, 36, 64, 3, 3, 1, False, 1, 256, ldx=40)  # ragged 32-chunk, padded pixel stride
    check(3, 6, 5, 16, 32, 3, 3, 3, True, 1, 192)             # dilation wider than half the image, dgrad, bf16
    print("halo model OK")
