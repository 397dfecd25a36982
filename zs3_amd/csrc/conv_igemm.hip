// Implicit-GEMM convolution for gfx950 (MI355X): forward and data-gradient.
//
//   y[m, co] = sum_{kh,kw,ci} x[gather(m,kh,kw), ci] * w[co, kh, kw, ci]        m = (n, ho, wo)
//
// GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin.  Activations are NHWC fp32 in HBM
// (channel-contiguous => the K axis of one filter tap is a contiguous 128-byte run per pixel, so
// the A gather is coalesced); weights arrive pre-split as two bf16 planes [Cout][K] (zs3_prep_weight).
// The A tile is split into bf16 hi/lo while it is staged into LDS, and every 32x32x16 MFMA is issued
// three times (lo*hi, hi*lo, hi*hi) into one fp32 accumulator ("bf16x3", common.h) -- fp32-class
// accuracy on the bf16 matrix cores.  PREC=1 issues only hi*hi (plain bf16 inputs).
//
// Block = 256 threads = 4 waves (2x2), block tile BM x BN x 32, wave tile (BM/2)x(BN/2) built from
// 32x32 MFMA tiles.  LDS rows are [32 hi | 32 lo | 8 pad] bf16 = 144 B: the 9-slot stride makes the
// ds_read_b128 fragment reads conflict-free.  Two LDS stages, one barrier per K step; the global
// loads of step k+1 are in flight while step k is multiplied.
//
// The same kernel computes dgrad: rows are dx pixels, the gather walks dy with the transposed
// stride relation, and the weight planes are the [Cin][KH*KW*Cout] transposes from zs3_prep_weight.
//
// Replaces: every nn.Conv2d on the reference hot path (resnet.py:16-28,79,125-131; aspp.py:11-19,86,97;
// decoder.py:12,16,20,26) and nn.Linear of the GMMN (gmmn.py:18,33) as a 1x1 conv.
#include "common.h"
#include "zs3hip.h"

namespace {

struct ConvArgs {
  const float* x;
  const unsigned short* w_hi;
  const unsigned short* w_lo;
  float* y;
  const float* scale;
  const float* shift;
  const float* res;
  float* stat_partial;
  int N, H, W, Ho, Wo;
  int cin_pad, cin_valid, ldx;
  int KH, KW, stride, pad_h, pad_w, dil;
  int ncols, ldw, ldy, ldr, M;
  int act, accumulate, dgrad;
  float leak;
};

template <int BM, int BN, int PREC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
  constexpr int ROW = 72;               // bf16 per LDS row (144 B)
  constexpr int RA = BM / 64, RB = BN / 64;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int STAGE = (BM + BN) * ROW;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.ncols + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / ntn, nt = bid - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread staging coordinates
  const int srow = tid >> 2, kc = (tid & 3) * 8;
  const float* xrow[RA];
  int bh[RA], bw[RA];
  bool rvalid[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = m0 + srow + 64 * i;
    rvalid[i] = m < p.M;
    int mm = rvalid[i] ? m : 0;
    int hw = p.Ho * p.Wo;
    int n = mm / hw, rem = mm - n * hw;
    int oh = rem / p.Wo, ow = rem - oh * p.Wo;
    xrow[i] = p.x + (size_t)n * p.H * p.W * p.ldx;
    if (p.dgrad) {
      bh[i] = oh + p.pad_h;
      bw[i] = ow + p.pad_w;
    } else {
      bh[i] = oh * p.stride - p.pad_h;
      bw[i] = ow * p.stride - p.pad_w;
    }
  }
  const unsigned short* wrow_hi[RB];
  const unsigned short* wrow_lo[RB];
  bool cvalid[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    int col = n0 + srow + 64 * j;
    cvalid[j] = col < p.ncols;
    size_t off = (size_t)(cvalid[j] ? col : 0) * p.ldw + kc;
    wrow_hi[j] = p.w_hi + off;
    wrow_lo[j] = p.w_lo + off;
  }

  f32x4 areg[RA][2];
  u32x4 breg_hi[RB], breg_lo[RB];
  int kh = 0, kw = 0, c0 = 0, kofs = 0;

  auto load_tile = [&]() {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int hi, wi;
      bool ok = rvalid[i] && (c0 + kc < p.cin_valid);
      if (p.dgrad) {
        int th = bh[i] - kh * p.dil, tw = bw[i] - kw * p.dil;
        hi = th / p.stride;
        wi = tw / p.stride;
        ok = ok && th >= 0 && tw >= 0 && (hi * p.stride == th) && (wi * p.stride == tw);
      } else {
        hi = bh[i] + kh * p.dil;
        wi = bw[i] + kw * p.dil;
        ok = ok && hi >= 0 && wi >= 0;
      }
      ok = ok && hi < p.H && wi < p.W;
      if (ok) {
        const float* ptr = xrow[i] + ((size_t)hi * p.W + wi) * p.ldx + c0 + kc;
        areg[i][0] = *reinterpret_cast<const f32x4*>(ptr);
        areg[i][1] = *reinterpret_cast<const f32x4*>(ptr + 4);
      } else {
        areg[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        areg[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      if (cvalid[j]) {
        breg_hi[j] = *reinterpret_cast<const u32x4*>(wrow_hi[j] + kofs);
        if (PREC == 3) breg_lo[j] = *reinterpret_cast<const u32x4*>(wrow_lo[j] + kofs);
      } else {
        breg_hi[j] = u32x4{0u, 0u, 0u, 0u};
        breg_lo[j] = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto advance = [&]() {
    kofs += 32;
    c0 += 32;
    if (c0 == p.cin_pad) {
      c0 = 0;
      if (++kw == p.KW) {
        kw = 0;
        ++kh;
      }
    }
  };
  auto store_tile = [&](int stage) {
    unsigned short* As = smem + stage * STAGE;
    unsigned short* Bs = As + BM * ROW;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      u32x4 hi, lo;
      unsigned h, l;
      split_pair<PREC>(areg[i][0][0], areg[i][0][1], h, l); hi[0] = h; lo[0] = l;
      split_pair<PREC>(areg[i][0][2], areg[i][0][3], h, l); hi[1] = h; lo[1] = l;
      split_pair<PREC>(areg[i][1][0], areg[i][1][1], h, l); hi[2] = h; lo[2] = l;
      split_pair<PREC>(areg[i][1][2], areg[i][1][3], h, l); hi[3] = h; lo[3] = l;
      unsigned short* dst = As + (srow + 64 * i) * ROW + kc;
      *reinterpret_cast<u32x4*>(dst) = hi;
      if (PREC == 3) *reinterpret_cast<u32x4*>(dst + 32) = lo;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      unsigned short* dst = Bs + (srow + 64 * j) * ROW + kc;
      *reinterpret_cast<u32x4*>(dst) = breg_hi[j];
      if (PREC == 3) *reinterpret_cast<u32x4*>(dst + 32) = breg_lo[j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage) {
    const unsigned short* As = smem + stage * STAGE + (wm * (BM / 2) + (lane & 31)) * ROW + (lane >> 5) * 8;
    const unsigned short* Bs = smem + stage * STAGE + BM * ROW + (wn * (BN / 2) + (lane & 31)) * ROW + (lane >> 5) * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a_hi[TM], a_lo[TM], b_hi[TN], b_lo[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        a_hi[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * ROW + kk * 16);
        if (PREC == 3) a_lo[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * ROW + kk * 16 + 32);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        b_hi[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * ROW + kk * 16);
        if (PREC == 3) b_lo[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * ROW + kk * 16 + 32);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (PREC == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[i], b_hi[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_lo[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  const int KT = p.KH * p.KW * (p.cin_pad / 32);
  load_tile();
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < KT;
    if (more) {
      advance();
      load_tile();
    }
    compute(cur);
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue 1: per-channel partial sums of the raw conv output (BatchNorm batch statistics)
  if (p.stat_partial) {
    float* red = reinterpret_cast<float*>(smem);  // [wm][{sum,sumsq}][BN]; tiles are no longer read
    float s[TN], q[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s[j] = 0.f;
      q[j] = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r];
          s[j] += v;
          q[j] = fmaf(v, v, q[j]);
        }
      s[j] += __shfl_xor(s[j], 32, 64);
      q[j] += __shfl_xor(q[j], 32, 64);
      if (lane < 32) {
        red[(wm * 2 + 0) * BN + wn * (BN / 2) + j * 32 + lane] = s[j];
        red[(wm * 2 + 1) * BN + wn * (BN / 2) + j * 32 + lane] = q[j];
      }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
      int col = n0 + c;
      if (col < p.ncols) {
        p.stat_partial[((size_t)mt * 2 + 0) * p.ncols + col] = red[c] + red[2 * BN + c];
        p.stat_partial[((size_t)mt * 2 + 1) * p.ncols + col] = red[BN + c] + red[3 * BN + c];
      }
    }
  }

  // ---- epilogue 2: affine / residual / activation / store
  const bool affine = (p.scale != nullptr) || (p.shift != nullptr);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
    const bool cok = col < p.ncols;
    const float sc = (cok && p.scale) ? p.scale[col] : 1.f;
    const float sh = (cok && p.shift) ? p.shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (cok && row < p.M) {
          float v = acc[i][j][r];
          if (affine) v = fmaf(v, sc, sh);
          if (p.res) v += p.res[(size_t)row * p.ldr + col];
          if (p.act == 1) v = fmaxf(v, 0.f);
          else if (p.act == 2) v = v > 0.f ? v : v * p.leak;
          float* dst = p.y + (size_t)row * p.ldy + col;
          if (p.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
}

template <int BM, int BN>
int launch_cfg(const ConvArgs& a, int prec, hipStream_t st) {
  int mt = (a.M + BM - 1) / BM, nt = (a.ncols + BN - 1) / BN;
  dim3 grid(mt * nt), block(256);
  if (prec == 1)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 1>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 3>), grid, block, 0, st, a);
  return ZS3_LAUNCH_CHECK();
}

}  // namespace

extern "C" int zs3_conv_igemm_mtiles(int M, int ncols, int tile_cfg) {
  int bm = 128;
  if (tile_cfg == 0) {
    int bn = ncols > 64 ? 128 : 64;
    long blocks = (long)((M + 127) / 128) * ((ncols + bn - 1) / bn);
    if (blocks < 512) bm = 64;
  } else {
    bm = (tile_cfg == 3 || tile_cfg == 4) ? 64 : 128;
  }
  return (M + bm - 1) / bm;
}

extern "C" int zs3_conv_igemm(const float* x, const void* w_hi, const void* w_lo, float* y, const float* scale,
                              const float* shift, const float* res, float* stat_partial, int N, int H, int W,
                              int Ho, int Wo, int cin_pad, int cin_valid, int ldx, int KH, int KW, int stride,
                              int pad_h, int pad_w, int dil, int ncols, int ldy, int ldr, int act, float leak,
                              int accumulate, int dgrad, int prec, int tile_cfg, void* stream) {
  if (cin_pad % 32 != 0 || cin_valid % 8 != 0 || ldx % 4 != 0 || (prec != 1 && prec != 3)) return -1;
  if (((uintptr_t)x & 15) || ((uintptr_t)w_hi & 15) || ((uintptr_t)w_lo & 15)) return -2;
  ConvArgs a;
  a.x = x; a.w_hi = (const unsigned short*)w_hi; a.w_lo = (const unsigned short*)w_lo; a.y = y;
  a.scale = scale; a.shift = shift; a.res = res; a.stat_partial = stat_partial;
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.cin_pad = cin_pad; a.cin_valid = cin_valid; a.ldx = ldx;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.dil = dil;
  a.ncols = ncols; a.ldw = KH * KW * cin_pad; a.ldy = ldy; a.ldr = ldr; a.M = N * Ho * Wo;
  a.act = act; a.accumulate = accumulate; a.dgrad = dgrad; a.leak = leak;
  if (a.M <= 0 || ncols <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int cfg = tile_cfg;
  if (cfg == 0) {
    int bn = ncols > 64 ? 128 : 64;
    long blocks = (long)((a.M + 127) / 128) * ((ncols + bn - 1) / bn);
    bool small = blocks < 512;
    cfg = bn == 128 ? (small ? 3 : 1) : (small ? 4 : 2);
  }
  switch (cfg) {
    case 1: return launch_cfg<128, 128>(a, prec, st);
    case 2: return launch_cfg<128, 64>(a, prec, st);
    case 3: return launch_cfg<64, 128>(a, prec, st);
    case 4: return launch_cfg<64, 64>(a, prec, st);
  }
  return -3;
}
