// Implicit-GEMM convolution for gfx950 (MI355X): forward and data-gradient.
//
//   y[m, co] = sum_{kh,kw,ci} x[gather(m,kh,kw), ci] * w[co, kh, kw, ci]        m = (n, ho, wo)
//
// GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin.  Activations are NHWC fp32 in HBM
// (channel-contiguous => the K axis of one filter tap is a contiguous 128-byte run per pixel, so
// the A gather is coalesced); weights arrive pre-split as bf16 hi/lo, interleaved per 32-wide K chunk
// ([Cout][K/32][{hi,lo}][32], zs3_prep_weight) so that a (row, chunk) is one 128-byte line as well.
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
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"
#include "zs3hip.h"

namespace {

// A16: x is stored as bf16 (plain-bf16 products only): written to LDS as it arrives.  KS = 64 (A16 only): a K step is 64 channels
// -- the LDS row's lo half holds channels 32..63 instead of a lo plane, a lane moves 16 bytes = 8 channels per row and only the
// hi halves of two weight chunks -- so one barrier covers twice the MFMAs (the plain-bf16 K loop is latency-, not MFMA-bound).
// Waves per SIMD the register allocation must leave room for (second __launch_bounds__ argument).  A 256-thread workgroup is one
// wave per SIMD, so this is the number of workgroups a CU can hold: with short K loops (the 1x1 layers: 4-16 K steps) the only
// thing that hides a tile's load latency and epilogue is ANOTHER workgroup's K loop.  hipcc left to itself allocates 136 registers
// for the 64 x 64 tile (3 per CU) and 276 for the 128 x 128 tile on bf16-stored input with 64-channel steps (1 per CU).
#ifndef ZS3_IGEMM_WPE64
#define ZS3_IGEMM_WPE64 4
#endif
#ifndef ZS3_IGEMM_WPE128A16
#define ZS3_IGEMM_WPE128A16 2
#endif
constexpr int igemm_wpe(int bm, int bn, int prec, int pipe, bool a16) {
  return (bm == 64 && bn == 64 && prec != 0) ? ZS3_IGEMM_WPE64 : (bm == 128 && bn == 128 && a16 && pipe <= 2) ? ZS3_IGEMM_WPE128A16 : 1;
}
template <int BM, int BN, int PREC, int PIPE, bool A16 = false, int KS = 32>
__global__ __launch_bounds__(256, igemm_wpe(BM, BN, PREC, PIPE, A16)) void conv_igemm_kernel(const ConvArgs p) {
  static_assert(!A16 || PREC == 1, "bf16-stored input: plain bf16 products only");
  static_assert(KS == 32 || (KS == 64 && A16), "64-channel K steps: bf16-stored input only");
  constexpr int CPL = KS / 8;                                    // channels per lane and row: 8 lanes walk a row's K step
  using XT = std::conditional_t<A16, bf16_t, float>;             // element type of x in memory
  using XV = std::conditional_t<A16, std::conditional_t<KS == 64, u32x4, u32x2>, f32x4>;   // CPL channels of it in registers
  const XT* const xbase = reinterpret_cast<const XT*>(p.x);
  const XT* const xzero = reinterpret_cast<const XT*>(p.zero);
  constexpr int ROW = 72;               // bf16 per LDS row (144 B)
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int STAGE = (BM + BN) * ROW;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.ncols + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / ntn, nt = bid - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread staging coordinates.  Every global load instruction covers whole 128-byte lines: 8 lanes
  // x 16 B walk one tile row (32 fp32 of A, or the 64 B hi + 64 B lo of B that zs3_prep_weight interleaves per
  // 32-wide K chunk), 8 rows per wave instruction.  Row group g = tid>>3 is bit-swapped (bits 0 <-> 2) so that the
  // lanes serviced together by one LDS write cycle land on rows 4 apart (144-byte rows => disjoint bank halves).
  const int q = tid & 7, tg = tid >> 3;
  const int srow = (tg & ~5) | ((tg & 1) << 2) | ((tg >> 2) & 1);   // 0..31
  constexpr int RA = BM / 32, RB = BN / 32;                          // rows of A / B staged per thread
  const XT* xrow[RA];
  int bh[RA], bw[RA];
  bool rvalid[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = m0 + srow + 32 * i;
    rvalid[i] = m < p.M;
    int mm = rvalid[i] ? m : 0;
    int hw = p.Ho * p.Wo;
    int n = mm / hw, rem = mm - n * hw;
    int oh = rem / p.Wo, ow = rem - oh * p.Wo;
    xrow[i] = xbase + (size_t)n * p.H * p.W * p.ldx;
    if (p.dgrad) {
      bh[i] = oh + p.pad_h;
      bw[i] = ow + p.pad_w;
    } else {
      bh[i] = oh * p.stride - p.pad_h;
      bw[i] = ow * p.stride - p.pad_w;
    }
  }
  const unsigned short* wrow[RB];
  int wstep[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    int col = n0 + srow + 32 * j;
    bool ok = col < p.ncols;
    // (KS = 64: the hi halves of two consecutive 32-wide chunks: piece q & 3 of chunk q >> 2)
    wrow[j] = ok ? p.w_pk + (size_t)col * (2 * p.ldw) + (KS == 64 ? (q >> 2) * 64 + (q & 3) * 8 : q * 8)
                 : reinterpret_cast<const unsigned short*>(p.zero);
    wstep[j] = ok ? 1 : 0;   // masked columns keep re-reading the zero page
  }

  struct Stage {
    XV areg[RA];
    u32x4 breg[RB];
  };
  // ---- filter taps that read nothing but padding for every row of this tile are skipped (as in conv_igemm_dma_kernel below: the K
  // loop is tap-major, a dead tap is a run of cin_pad / KS steps multiplying zeros -- bit-identical without them).  Only asked for
  // dilations >= 4: ASPP's d = 18 branch on a 33 x 33 map lands here in the 2-byte mode (the LDS-DMA kernel moves fp32 rows, the
  // strip of d = 18 does not fit the LDS), and a 64-row tile is two image rows: 5.7 of its 9 taps are live on average.
  const int T = p.KH * p.KW;
  unsigned tapmask = T <= 32 ? (T == 32 ? 0xFFFFFFFFu : (1u << T) - 1u) : 0u;
  if (T > 1 && T <= 32 && p.dil >= 4) {
    __shared__ unsigned s_tapmask;
    if (tid == 0) s_tapmask = 0u;
    unsigned mk = 0u;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      if (!rvalid[i]) continue;
      for (int t = 0; t < T; ++t) {
        const int th_ = t / p.KW, tw_ = t - th_ * p.KW;
        int hi, wi;
        bool ok;
        if (p.dgrad) {
          const int th = bh[i] - th_ * p.dil, tw = bw[i] - tw_ * p.dil;
          const int smask = (1 << p.stride_log2) - 1;
          hi = th >> p.stride_log2;
          wi = tw >> p.stride_log2;
          ok = ((th | tw) >= 0) && (((th | tw) & smask) == 0);
        } else {
          hi = bh[i] + th_ * p.dil;
          wi = bw[i] + tw_ * p.dil;
          ok = (hi | wi) >= 0;
        }
        if (ok && hi < p.H && wi < p.W) mk |= 1u << t;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mk |= (unsigned)__shfl_xor((int)mk, o, 64);
    __syncthreads();
    if (lane == 0) atomicOr(&s_tapmask, mk);
    __syncthreads();
    tapmask = s_tapmask;
    if (tapmask == 0u) tapmask = 1u;
  }
  int tcur = T <= 32 ? __builtin_ctz(tapmask) : 0;
  int kh = tcur / p.KW, kw = tcur - kh * p.KW, c0 = 0, kofs = tcur * p.cin_pad;

  // Branch-free tile loads: masked lanes read the zero page, so the loop body is straight-line code.  The
  // per-row gather address (bounds checks, 64-bit pointer) is recomputed only when a new filter tap
  // starts (c0 == 0, a wave-uniform test); inside a tap the pointer just advances by 32 channels.
  const XT* abase[RA];
  int astep[RA];
  auto load_tile = [&](Stage& S) {
    if (c0 == 0) {
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        int hi, wi;
        bool ok = rvalid[i];
        if (p.dgrad) {
          const int th = bh[i] - kh * p.dil, tw = bw[i] - kw * p.dil;
          const int mask = (1 << p.stride_log2) - 1;
          hi = th >> p.stride_log2;
          wi = tw >> p.stride_log2;
          ok = ok && ((th | tw) >= 0) && (((th | tw) & mask) == 0);
        } else {
          hi = bh[i] + kh * p.dil;
          wi = bw[i] + kw * p.dil;
          ok = ok && ((hi | wi) >= 0);
        }
        ok = ok && hi < p.H && wi < p.W;
        abase[i] = ok ? xrow[i] + ((hi * p.W + wi) * p.ldx + q * CPL) : xzero;
        astep[i] = ok ? 1 : 0;
      }
    }
    const bool cok = c0 + q * CPL < p.cin_valid;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const XT* ptr = cok ? abase[i] + c0 * astep[i] : xzero;
      S.areg[i] = *reinterpret_cast<const XV*>(ptr);
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) S.breg[j] = *reinterpret_cast<const u32x4*>(wrow[j] + 2 * kofs * wstep[j]);
  };
  auto advance = [&]() {
    kofs += KS;
    c0 += KS;
    if (c0 == p.cin_pad) {   // next live tap
      c0 = 0;
      ++tcur;
      if (T <= 32) {
        const unsigned rest = tcur < 32 ? tapmask >> tcur : 0u;
        tcur += rest ? __builtin_ctz(rest) : 0;
      }
      kh = tcur / p.KW;
      kw = tcur - kh * p.KW;
      kofs = tcur * p.cin_pad;
    }
  };
  auto store_tile = [&](Stage& S, int stage) {
    unsigned short* As = smem + stage * STAGE;
    unsigned short* Bs = As + BM * ROW;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      if constexpr (PREC == 0) {   // exact fp32 (test mode): the row holds the 32 raw floats of the K step
        *reinterpret_cast<f32x4*>(As + (srow + 32 * i) * ROW + q * 8) = S.areg[i];
        continue;
      }
      if constexpr (A16) {         // already bf16: the lane's channels go to LDS as they came
        *reinterpret_cast<XV*>(As + (srow + 32 * i) * ROW + q * CPL) = S.areg[i];
        continue;
      }
      u32x2 hi, lo;
      unsigned h, l;
      split_pair<PREC>(S.areg[i][0], S.areg[i][1], h, l); hi[0] = h; lo[0] = l;
      split_pair<PREC>(S.areg[i][2], S.areg[i][3], h, l); hi[1] = h; lo[1] = l;
      unsigned short* dst = As + (srow + 32 * i) * ROW + q * 4;
      *reinterpret_cast<u32x2*>(dst) = hi;
      if (PREC >= 3) *reinterpret_cast<u32x2*>(dst + 32) = lo;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j)
      *reinterpret_cast<u32x4*>(Bs + (srow + 32 * j) * ROW + (q & 3) * 8 + (q >> 2) * 32) = S.breg[j];   // = q * 8: K order
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage) {
    if constexpr (PREC == 0) {
      // v_mfma_f32_32x32x2_f32: lanes 0-31 supply k = 0, lanes 32-63 k = 1 of each instruction.  The reduction order inside a
      // K step is free as long as A and B agree: instruction t of sub-step kk multiplies k = 16 kk + 8 (lane >> 5) + t, so a
      // lane reads 8 consecutive floats (two ds_read_b128) of its row per sub-step.
      const unsigned short* A0 = smem + stage * STAGE + (wm * (BM / 2) + (lane & 31)) * ROW + (lane >> 5) * 16;
      const unsigned short* B0 = smem + stage * STAGE + BM * ROW + (wn * (BN / 2) + (lane & 31)) * ROW + (lane >> 5) * 16;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        f32x4 a[TM][2], b[TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) a[i][h] = *reinterpret_cast<const f32x4*>(A0 + i * 32 * ROW + kk * 32 + h * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int h = 0; h < 2; ++h) b[j][h] = *reinterpret_cast<const f32x4*>(B0 + j * 32 * ROW + kk * 32 + h * 8);
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t >> 2][t & 3], b[j][t >> 2][t & 3], acc[i][j], 0, 0, 0);
      }
      return;
    }
    const unsigned short* As = smem + stage * STAGE + (wm * (BM / 2) + (lane & 31)) * ROW + (lane >> 5) * 8;
    const unsigned short* Bs = smem + stage * STAGE + BM * ROW + (wn * (BN / 2) + (lane & 31)) * ROW + (lane >> 5) * 8;
#pragma unroll
    for (int kk = 0; kk < KS / 16; ++kk) {
      bf16x8 a_hi[TM], a_lo[TM], b_hi[TN], b_lo[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        a_hi[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * ROW + kk * 16);
        if (PREC >= 3) a_lo[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * ROW + kk * 16 + 32);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        b_hi[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * ROW + kk * 16);
        if (PREC >= 3) b_lo[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * ROW + kk * 16 + 32);
      }
      if (PREC >= 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = mfma16<PREC>(a_lo[i], b_hi[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = mfma16<PREC>(a_hi[i], b_lo[j], acc[i][j]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = mfma16<PREC>(a_hi[i], b_hi[j], acc[i][j]);
    }
  };

  const int KT = (T <= 32 ? __builtin_popcount(tapmask) : T) * (p.cin_pad / KS);
  if (PIPE == 1) {
    // one register stage: loads of step k+1 fly during the MFMAs of step k
    Stage s0;
    load_tile(s0);
    store_tile(s0, 0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int cur = kt & 1;
      const bool more = kt + 1 < KT;
      if (more) {
        advance();
        load_tile(s0);
      }
      compute(cur);
      if (more) store_tile(s0, cur ^ 1);
      __syncthreads();
    }
  } else {
    // two register stages: a tile's global loads are issued two K steps before they are written to LDS
    Stage s0, s1;
    load_tile(s0);
    if (KT > 1) {
      advance();
      load_tile(s1);
    }
    store_tile(s0, 0);
    if (KT > 2) {
      advance();
      load_tile(s0);
    }
    __syncthreads();
    // steady state handles two K steps per trip (no exit in the middle: the accumulators keep one register
    // assignment across the loop); an odd last step is peeled
    int kt = 0;
    for (; kt + 1 < KT; kt += 2) {
      compute(0);
      store_tile(s1, 1);
      if (kt + 3 < KT) {
        advance();
        load_tile(s1);
      }
      __syncthreads();
      compute(1);
      if (kt + 2 < KT) store_tile(s0, 0);
      if (kt + 4 < KT) {
        advance();
        load_tile(s0);
      }
      __syncthreads();
    }
    if (kt < KT) compute(0);
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) unscale_acc<PREC>(acc[i][j]);   // (f16x3: the forward plane carries 2^6 w)

  // ---- epilogue 1: per-channel partial sums of the raw conv output (BatchNorm batch statistics)
  if (p.stat_partial) {
    // The sums are staged in the first bytes of LDS stage 0, and the multi-stage loops peel their last K step(s) WITHOUT a closing
    // barrier: when the last step's operands sit in stage 0 (an odd step count: the 7-tap stem; with 64-channel steps every
    // 64-channel 1x1 / 3x3 layer of layer 1) a fast wave would overwrite operands a slow wave is still multiplying.  Round 4 found
    // it twice: the stem's output came out as zeros under the three-stage loop, and the 2-byte mode's step was not
    // bit-reproducible (tools/probe/determinism.py; NaN-poisoned allocations, tools/probe/nanfill.py, put the first bad tensor at
    // layer1.0.conv1, a ONE-step launch).  One barrier per tile; the launch-side routing of odd step counts to the one-stage loop
    // that stood in for it is gone.
    __syncthreads();
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

  // ---- epilogue 2: affine / residual / activation / store.  The accumulator tile goes through LDS so that
  // global stores (and residual / accumulate loads) are whole 512-byte rows as dwordx4 per lane instead of 64
  // scattered 4-byte stores per lane (the MFMA C layout gives each lane one column of 16 rows).
  __syncthreads();  // all waves are done with the operand tiles (and with `red`)
  {
    constexpr int LDC = BN + 4;  // floats; +4 keeps rows 16-byte aligned and spreads the half-wave row offset over banks
    static_assert(BM * LDC * 4 <= 2 * STAGE * 2, "output tile must fit in the operand LDS");
    float* ctile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          ctile[row * LDC + wn * (BN / 2) + j * 32 + (lane & 31)] = acc[i][j][r];
        }
    __syncthreads();
    const bool affine = (p.scale != nullptr) || (p.shift != nullptr);
    constexpr int C4 = BN / 4;                 // float4 columns per row
    constexpr int RPP = 256 / C4;              // rows per pass
    const int c4 = tid % C4, r0 = tid / C4;
    const int col = n0 + c4 * 4;
    const bool vec = ((p.ldy & 3) == 0) && ((p.ncols & 3) == 0) && (!p.res || (p.ldr & 3) == 0);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col + e < p.ncols) {
        if (p.scale) sc[e] = p.scale[col + e];
        if (p.shift) sh[e] = p.shift[col + e];
      }
    f32x4 bs_s = {0.f, 0.f, 0.f, 0.f}, bs_q = {0.f, 0.f, 0.f, 0.f};
    store_tile_rows<RPP>(p, ctile, LDC, m0, BM, col, c4, r0, sc, sh, affine, vec, bs_s, bs_q);
    if (p.bs_partial) finish_bwd_stats<BN, RPP, 256>(p, ctile, tid, c4, r0, mt, n0, bs_s, bs_q);
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant of the wave-specialised 256x128x32 kernel (tile_cfg 31).
//
// Producers (waves 4-7) never touch the data: every tile row is fetched with global_load_lds_dwordx4, which
// moves 64 lanes x 16 B straight from L2/HBM into LDS (wave-uniform LDS base + lane*16; the *global* address is
// per lane).  The A tile therefore sits in LDS as raw fp32 (256 rows x 128 B) and the B tile as the packed
// bf16 {hi,lo} lines of zs3_prep_weight (128 rows x 128 B); with no staging registers the prefetch depth is the
// LDS ring: three 48 KB stages, tiles k+1 and k+2 in flight while tile k is multiplied, one s_barrier per K
// step, and the producers only wait on a counted vmcnt (never 0 inside the loop).
// Rows are 128 B apart, so the 16-byte chunks of row r are stored XOR-swizzled (chunk c at slot
// c ^ ((r>>1)&7)): the producer applies it to the lane's global source address, the consumer to its read
// address, and the ds_read_b128 fragment reads of 16 different rows hit 16 different bank quads.
// Consumers (waves 0-3, one 64x128 wave tile each, so every A value is converted exactly once) read 8 fp32 of
// their row, split them into bf16 hi/lo in registers (2 VALU per MFMA, hidden under the 32-cycle MFMAs) and
// issue the three bf16 MFMAs per product.

template <int PREC>
__global__ __launch_bounds__(512) void conv_igemm_dma_kernel(const ConvArgs p) {
  constexpr int BM = 256, BN = 128, TM = 2, TN = 4, NST = 3;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int tid_outer = tid, lane_outer = lane;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int ntn = (p.ncols + BN - 1) / BN;
  const int wm = wave & 3;
#ifdef ZS3_CONV_TIMING
  const long t_kernel0 = __builtin_readcyclecounter();
  long t_loop0 = 0, t_loop1 = 0;
#endif
  // ---- work of this block: one output tile, all of its K steps.  (Round 2 also ran this kernel as a stream-K launch -- 256
  // persistent blocks sharing the (tile, K step) iterations, partial tiles through a per-stream workspace: -8 ... -28 % per
  // isolated launch, +0.5 ms and occasional multi-ms stalls inside the training step, where the CUs a 138-tile launch
  // leaves idle are taken by the weight-gradient streams.  The layers it applied to (3x3 of layer 3, ASPP) moved to the
  // strip-resident kernel in round 3 and the path was removed.)
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / ntn, nt = tile - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  // ---- filter taps that read nothing but padding for EVERY row of this tile are skipped (round 5).  The K loop runs tap-major,
  // so a dead tap is a contiguous run of cin_pad / 32 K steps whose A rows all come from the zero page: 3x3 at dilation 18 on a
  // 33 x 33 map -- ASPP's last branch -- has its taps above (below) the centre row dead for every output row >= 15 (<= 17), and a
  // tile is 256 consecutive pixels = 7.8 image rows: 5.7 of 9 taps live on average (6.8 at dilation 12, 8.5 at dilation 6).
  // Adding exact zeros to an accumulator changes nothing, so the result is bit-identical.  Every wave derives the same mask
  // (lane l looks at rows l, l + 64, l + 128, l + 192; an OR across the wave), hence the same K-step count -- no LDS, no barrier.
  const int T = p.KH * p.KW;
  unsigned tapmask = T <= 32 ? (T == 32 ? 0xFFFFFFFFu : (1u << T) - 1u) : 0u;
  if (T > 1 && T <= 32) {
    unsigned mk = 0u;
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int q = 0; q < BM / 64; ++q) {
      const int m = m0 + q * 64 + lane;
      if (m < p.M) {
        const int n = m / hw, rem = m - n * hw;
        const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
        for (int t = 0; t < T; ++t) {
          const int th_ = t / p.KW, tw_ = t - th_ * p.KW;
          int hi, wi;
          bool ok;
          if (p.dgrad) {
            const int th = oh + p.pad_h - th_ * p.dil, tw = ow + p.pad_w - tw_ * p.dil;
            const int smask = (1 << p.stride_log2) - 1;
            hi = th >> p.stride_log2;
            wi = tw >> p.stride_log2;
            ok = ((th | tw) >= 0) && (((th | tw) & smask) == 0);
          } else {
            hi = oh * p.stride - p.pad_h + th_ * p.dil;
            wi = ow * p.stride - p.pad_w + tw_ * p.dil;
            ok = (hi | wi) >= 0;
          }
          if (ok && hi < p.H && wi < p.W) mk |= 1u << t;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mk |= (unsigned)__shfl_xor((int)mk, o, 64);
    tapmask = (unsigned)__builtin_amdgcn_readfirstlane((int)mk);
    if (tapmask == 0u) tapmask = 1u;     // (cannot happen for a tile with a row inside M; keeps the loop well-formed)
  }
  const int KT = (T <= 32 ? __builtin_popcount(tapmask) : T) * (p.cin_pad / 32);   // K steps of this tile
  // ---- the tile's fused epilogue.  Runs in both wave roles with the same barrier sequence; only the consumers hold
  // accumulators (`acc` is a dummy for the producers).
  auto finish_segment = [&](auto role, auto& acc) {
    constexpr bool IS_PRODUCER = decltype(role)::value;
    // the thread index is re-read through an opaque asm: everything the epilogue derives from it (column offsets, scale /
    // shift vectors, store addresses) would otherwise be hoisted out of the stream-K segment loop and kept alive across the
    // K loop, where the accumulators already fill the register file (-> scratch spills, and a kernel with scratch pays a
    // ~1 ms dispatch penalty on this runtime)
    int tid = tid_outer, lane = lane_outer;
    asm volatile("" : "+v"(tid), "+v"(lane));
  constexpr int LDC = BN + 4;
  static_assert((BM / 2) * LDC * 4 <= NST * STAGE_BYTES, "half output tile must fit in the operand LDS");
  float* ctile = reinterpret_cast<float*>(dsm);
  // ---- epilogue (consumers hold the accumulators; every DMA has landed and been consumed)
  if (p.stat_partial) {
    float* red = reinterpret_cast<float*>(dsm);
    if constexpr (!IS_PRODUCER) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float s = 0.f, q2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][j][r];
            s += v;
            q2 = fmaf(v, v, q2);
          }
        s += __shfl_xor(s, 32, 64);
        q2 += __shfl_xor(q2, 32, 64);
        if (lane < 32) {
          red[(wm * 2 + 0) * BN + j * 32 + lane] = s;
          red[(wm * 2 + 1) * BN + j * 32 + lane] = q2;
        }
      }
    }
    __syncthreads();
    if (tid < BN) {
      int col = n0 + tid;
      if (col < p.ncols) {
        p.stat_partial[((size_t)mt * 2 + 0) * p.ncols + col] =
            (red[tid] + red[2 * BN + tid]) + (red[4 * BN + tid] + red[6 * BN + tid]);
        p.stat_partial[((size_t)mt * 2 + 1) * p.ncols + col] =
            (red[BN + tid] + red[3 * BN + tid]) + (red[5 * BN + tid] + red[7 * BN + tid]);
      }
    }
  }
  const bool affine = (p.scale != nullptr) || (p.shift != nullptr);
  constexpr int C4 = BN / 4, RPP = 512 / C4;
  const int c4 = tid % C4, r0 = tid / C4;
  const int col = n0 + c4 * 4;
  const bool vec = ((p.ldy & 3) == 0) && ((p.ncols & 3) == 0) && (!p.res || (p.ldr & 3) == 0);
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (col + e < p.ncols) {
      if (p.scale) sc[e] = p.scale[col + e];
      if (p.shift) sh[e] = p.shift[col + e];
    }
  f32x4 bs_s = {0.f, 0.f, 0.f, 0.f}, bs_q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    if constexpr (!IS_PRODUCER) if ((wm >> 1) == half) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (wm & 1) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            ctile[row * LDC + j * 32 + (lane & 31)] = acc[i][j][r];
          }
    }
    __syncthreads();
    store_tile_rows<RPP>(p, ctile, LDC, m0 + half * (BM / 2), BM / 2, col, c4, r0, sc, sh, affine, vec, bs_s, bs_q);
  }
  if (p.bs_partial) finish_bwd_stats<BN, RPP, 512>(p, ctile, tid, c4, r0, mt, n0, bs_s, bs_q);
#ifdef ZS3_CONV_TIMING
  if (p.act == 99 && blockIdx.x == 0 && tid == 0) {
    long* o = reinterpret_cast<long*>(const_cast<float*>(p.res)) + 24;
    o[0] = t_loop0 - t_kernel0;                       // prologue (consumer wave 0)
    o[1] = __builtin_readcyclecounter() - t_loop1;    // epilogue
    o[2] = t_loop1 - t_loop0;                         // K loop
  }
#endif
  };
  if (producer) {
    int no_acc = 0;
    {
    const int pw = wave - 4;
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));
      const int lrow = lane_s >> 3, slot = lane_s & 7;
      constexpr int RA = 8, RB = 4;   // 8-row groups of A / B fetched per producer wave and K step
      // group g of A holds tile rows 8g..8g+7 (g = 8 pw + i); the swizzle of row r is (r>>1)&7 = ((i&1)<<2) | (lane>>4)
      const int chunk_even = slot ^ (lane >> 4), chunk_odd = slot ^ (4 | (lane >> 4));
      const float* xrow[RA];
      int bh[RA], bw[RA];
      bool rvalid[RA];
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        int m = m0 + 64 * pw + 8 * i + lrow;
        rvalid[i] = m < p.M;
        int mm = rvalid[i] ? m : 0;
        int hw = p.Ho * p.Wo;
        int n = mm / hw, rem = mm - n * hw;
        int oh = rem / p.Wo, ow = rem - oh * p.Wo;
        xrow[i] = p.x + (size_t)n * p.H * p.W * p.ldx + ((i & 1) ? chunk_odd : chunk_even) * 4;
        if (p.dgrad) {
          bh[i] = oh + p.pad_h;
          bw[i] = ow + p.pad_w;
        } else {
          bh[i] = oh * p.stride - p.pad_h;
          bw[i] = ow * p.stride - p.pad_w;
        }
      }
      const unsigned short* wrow[RB];
      int wstep[RB];
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        int col = n0 + 32 * pw + 8 * j + lrow;
        bool ok = col < p.ncols;
        wrow[j] = ok ? p.w_pk + (size_t)col * (2 * p.ldw) + ((j & 1) ? chunk_odd : chunk_even) * 8
                     : reinterpret_cast<const unsigned short*>(p.zero);
        wstep[j] = ok ? 1 : 0;
      }
      // current filter tap (live taps only, ascending) / channel offset inside it / offset in the K = taps x cin_pad axis of the weights
      int tcur = T <= 32 ? __builtin_ctz(tapmask) : 0;
      int kh = tcur / p.KW, kw = tcur - kh * p.KW, c0 = 0, kofs = tcur * p.cin_pad;
      const float* abase[RA];
      int astep[RA];
      auto tap_addresses = [&]() {   // per-row gather base of the current filter tap (called when a tap starts)
        {
#pragma unroll
          for (int i = 0; i < RA; ++i) {
            int hi, wi;
            bool ok = rvalid[i];
            if (p.dgrad) {
              const int th = bh[i] - kh * p.dil, tw = bw[i] - kw * p.dil;
              const int mask = (1 << p.stride_log2) - 1;
              hi = th >> p.stride_log2;
              wi = tw >> p.stride_log2;
              ok = ok && ((th | tw) >= 0) && (((th | tw) & mask) == 0);
            } else {
              hi = bh[i] + kh * p.dil;
              wi = bw[i] + kw * p.dil;
              ok = ok && ((hi | wi) >= 0);
            }
            ok = ok && hi < p.H && wi < p.W;
            abase[i] = ok ? xrow[i] + (hi * p.W + wi) * p.ldx : p.zero;
            astep[i] = ok ? 1 : 0;
          }
        }
      };
      auto advance = [&]() {
        kofs += 32;
        c0 += 32;
        if (c0 == p.cin_pad) {   // next LIVE tap (scalar code; past the last one the values are never used)
          c0 = 0;
          ++tcur;
          if (T <= 32) {
            const unsigned rest = tcur < 32 ? tapmask >> tcur : 0u;
            tcur += rest ? __builtin_ctz(rest) : 0;
          }
          kh = tcur / p.KW;
          kw = tcur - kh * p.KW;
          kofs = tcur * p.cin_pad;
        }
      };
      // (A register-staged producer -- global_load_dwordx4 + ds_write_b128 of the same bytes -- was measured against this
      // LDS-DMA form: 2700 vs 2100 producer cycles per K step; both sit on the ~35 B/clk/CU the L2 delivers.)
      bool fresh = true;   // first tile of the segment: it may start in the middle of a filter tap
      auto issue_tile = [&](int stage) {
        if (c0 == 0 || fresh) tap_addresses();
        fresh = false;
        unsigned char* sA = dsm + stage * STAGE_BYTES + pw * (8 * 1024);
        unsigned char* sB = dsm + stage * STAGE_BYTES + A_BYTES + pw * (4 * 1024);
#ifdef ZS3_DMA_ZEROSRC   // probe: every lane re-reads the (L1-resident) zero page -> separates issue cost from L2 bandwidth
        if (true) {
#pragma unroll
          for (int i = 0; i < RA; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(p.zero + (lane & 63) * 4), (lds_void_t*)(sA + i * 1024), 16, 0, 0);
#pragma unroll
          for (int j = 0; j < RB; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(p.zero + (lane & 63) * 4), (lds_void_t*)(sB + j * 1024), 16, 0, 0);
          advance();
          return;
        }
#endif
        if (p.cin_valid == p.cin_pad) {
#pragma unroll
          for (int i = 0; i < RA; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(abase[i] + c0 * astep[i]), (lds_void_t*)(sA + i * 1024), 16, 0, 0);
        } else {   // ragged last channel chunk of a tap: lanes past cin_valid fetch the zero page
          const int rem = p.cin_valid - c0;
          const bool ok_even = chunk_even * 4 < rem, ok_odd = chunk_odd * 4 < rem;
#pragma unroll
          for (int i = 0; i < RA; ++i) {
            const float* src = ((i & 1) ? ok_odd : ok_even) ? abase[i] + c0 * astep[i] : p.zero;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(sA + i * 1024), 16, 0, 0);
          }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j)
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(wrow[j] + 2 * kofs * wstep[j]), (lds_void_t*)(sB + j * 1024), 16, 0,
                                           0);
        advance();
      };
      issue_tile(0);
      if (KT > 1) {
        issue_tile(1);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();  // tile 0 has landed
      int st2 = 2;                   // ring slot of tile kt + 2
#ifdef ZS3_CONV_TIMING
      long t_issue = 0, t_wait = 0, t_bar = 0;
#define ZS3_T(v) { long t_ = __builtin_readcyclecounter(); v += t_ - t_last; t_last = t_; }
      long t_last = __builtin_readcyclecounter();
#else
#define ZS3_T(v)
#endif
      for (int kt = 0; kt < KT; ++kt) {
        if (kt + 2 < KT) {
          issue_tile(st2);
          st2 = st2 == NST - 1 ? 0 : st2 + 1;
          ZS3_T(t_issue)
          asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // tile kt+1 has landed, tile kt+2 stays in flight
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ZS3_T(t_wait)
        __builtin_amdgcn_s_barrier();
        ZS3_T(t_bar)
      }
#ifdef ZS3_CONV_TIMING
      if (p.act == 99 && blockIdx.x == 0 && lane == 0) {
        long* o = reinterpret_cast<long*>(const_cast<float*>(p.res)) + wave * 3;
        o[0] = t_issue; o[1] = t_wait; o[2] = t_bar;
      }
#endif
      finish_segment(std::true_type{}, no_acc);
    }
  } else {
    f32x16 acc[TM][TN];
    {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      int lane_s = lane;   // re-read per segment (see finish_segment): keeps the swizzled fragment offsets out of the
      asm volatile("" : "+v"(lane_s));   // set of values that live across the whole segment loop
      const int lr = lane_s & 31, sw = (lr >> 1) & 7, jh = lane_s >> 5;
      int offA0[2], offA1[2], offBh[2], offBl[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int j4 = jh + 2 * kk;              // which 8-wide K group of the 32-chunk this lane's fragment holds
        offA0[kk] = ((2 * j4) ^ sw) * 16;        // fp32 chunks 2*j4, 2*j4+1
        offA1[kk] = ((2 * j4 + 1) ^ sw) * 16;
        offBh[kk] = (j4 ^ sw) * 16;              // bf16 hi chunk j4, lo chunk 4 + j4
        offBl[kk] = ((4 + j4) ^ sw) * 16;
      }
      const int rowA = (wm * 64 + lr) * 128, rowB = A_BYTES + lr * 128;
      // One K step = 4 sub-steps (kk, i) of 12 MFMAs each (3 bf16 products x 4 column tiles; the three updates of one
      // accumulator are 4 MFMAs apart).  The slots between the MFMAs of sub-step s carry the LDS reads of the next raw
      // fp32 fragment / the next B fragments and the hi/lo split for sub-step s+1 (3 VALU per slot), so the conversion
      // work sits in the shadow of the 32-cycle MFMAs.  The K-step barrier sits before sub-step 3, which works from
      // registers only and meanwhile prefetches sub-step 0 of the next tile from the next ring slot.  hipcc's own
      // schedule hoists all LDS reads and conversions to the top and issues the 48 MFMAs as one clump (no overlap),
      // so every slot is pinned with sched_barrier(0).
      bf16x8 b_hi[2][TN], b_lo[2][TN];
      u32x4 uh[2], ul[2];
      f32x4 r0, r1;
      float ha = 0.f, hb = 0.f;
      const unsigned char* Ab;
      const unsigned char* Bb;
      auto set_stage = [&](int stage) {
        Ab = dsm + stage * STAGE_BYTES + rowA;
        Bb = dsm + stage * STAGE_BYTES + rowB;
      };
      auto read_a = [&](int kk, int i) {
        r0 = *reinterpret_cast<const f32x4*>(Ab + i * 4096 + offA0[kk]);
        r1 = *reinterpret_cast<const f32x4*>(Ab + i * 4096 + offA1[kk]);
      };
      auto read_b = [&](int kk, int j) {
        b_hi[kk][j] = *reinterpret_cast<const bf16x8*>(Bb + j * 4096 + offBh[kk]);
        if (PREC >= 3) b_lo[kk][j] = *reinterpret_cast<const bf16x8*>(Bb + j * 4096 + offBl[kk]);
      };
      auto split_half = [&](int buf, int hp) {   // hp = 2*q + phase: values 2q, 2q+1 of the 8-wide fragment
        const int q = hp >> 1;
        const float a = q == 0 ? r0[0] : q == 1 ? r0[2] : q == 2 ? r1[0] : r1[2];
        const float b = q == 0 ? r0[1] : q == 1 ? r0[3] : q == 2 ? r1[1] : r1[3];
        // two phases per pair, spread over the MFMA slots: hi = the packed conversion (+ for bf16 the two fp32 images of its
        // halves); lo = fp16: the two mixed-precision FMAs of common.h (1 + 2 VALU), bf16: two subtractions + the conversion (3 + 3)
        if ((hp & 1) == 0) {
          if constexpr (PREC == 4) {
            uh[buf][q] = cvt_pk_f16(a, b);
          } else {
            const unsigned h = cvt_pk_bf16(a, b);
            uh[buf][q] = h;
            ha = __uint_as_float(h << 16);
            hb = __uint_as_float(h & 0xFFFF0000u);
          }
        } else {
          ul[buf][q] = PREC == 4 ? f16_lo_pair(uh[buf][q], a, b) : PREC == 3 ? cvt_pk_bf16(a - ha, b - hb) : 0u;
        }
      };
      // sub-step s4 of the tile in the current ring slot; fillers prepare sub-step s4+1 (s4 == 3: sub-step 0 of the
      // tile in the slot set_stage() was last pointed at)
      auto substep = [&](int s4) {
        const int kk = s4 >> 1, i = s4 & 1, buf = s4 & 1;
        const int nkk = ((s4 + 1) & 3) >> 1, ni = (s4 + 1) & 1;
        const bf16x8 a_hi = __builtin_bit_cast(bf16x8, uh[buf]);
        const bf16x8 a_lo = __builtin_bit_cast(bf16x8, ul[buf]);
        if (PREC >= 3) {
#pragma unroll
          for (int g = 0; g < 12; ++g) {
            const int t = g >> 2, j = g & 3;
            acc[i][j] = mfma16<PREC>(t == 0 ? a_lo : a_hi, t == 1 ? b_lo[kk][j] : b_hi[kk][j], acc[i][j]);
#ifndef ZS3_DMA_ABLATE
#define ZS3_DMA_ABLATE 0
#endif
            if (!(ZS3_DMA_ABLATE & 2)) {
              if (g == 0) read_a(nkk, ni);
              if ((s4 == 0 || s4 == 3) && g >= 1 && g <= 4) read_b(s4 == 0 ? 1 : 0, g - 1);   // B of the next kk
            }
            if (!(ZS3_DMA_ABLATE & 1) && g >= 4) split_half(buf ^ 1, g - 4);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = mfma16<PREC>(a_hi, b_hi[kk][j], acc[i][j]);
            if (j == 0) read_a(nkk, ni);
            if (s4 == 0 || s4 == 3) read_b(s4 == 0 ? 1 : 0, j);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int hp = 0; hp < 8; ++hp) split_half(buf ^ 1, hp);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      __builtin_amdgcn_s_barrier();  // tile 0 has landed
      asm volatile("" ::: "memory");
      int st = 0;
      set_stage(0);
      read_a(0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) read_b(0, j);
#pragma unroll
      for (int hp = 0; hp < 8; ++hp) split_half(0, hp);
      __builtin_amdgcn_sched_barrier(0);
#ifdef ZS3_CONV_TIMING
      long t_work = 0, t_bar = 0;
      long t_last = __builtin_readcyclecounter();
      t_loop0 = t_last;
#endif
      for (int kt = 0; kt < KT; ++kt) {
        substep(0);
        substep(1);
        substep(2);
        st = st == NST - 1 ? 0 : st + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ZS3_T(t_work)
        __builtin_amdgcn_s_barrier();  // tile kt+1 has landed; nobody reads tile kt from LDS any more
        asm volatile("" ::: "memory");
        ZS3_T(t_bar)
        set_stage(st);
        substep(3);                    // after the last tile this prefetches stale (unused) data
      }
#ifdef ZS3_CONV_TIMING
      t_loop1 = __builtin_readcyclecounter();
      if (p.act == 99 && blockIdx.x == 0 && lane == 0) {
        long* o = reinterpret_cast<long*>(const_cast<float*>(p.res)) + wave * 3;
        o[0] = t_work; o[1] = 0; o[2] = t_bar;
      }
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) unscale_acc<PREC>(acc[i][j]);   // (f16x3: the forward plane carries 2^6 w)
        finish_segment(std::false_type{}, acc);
    }
  }
}

template <int PREC>
int launch_dma_prec(const ConvArgs& a, int grid, hipStream_t st) {
  constexpr int LDS_BYTES = 3 * (256 + 128) * 128;   // 144 KB of the CU's 160 KB
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_dma_kernel<PREC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return -4;
    configured = true;
  }
  hipLaunchKernelGGL((conv_igemm_dma_kernel<PREC>), dim3(grid), dim3(512), LDS_BYTES, st, a);
  return ZS3_LAUNCH_CHECK();
}

int launch_dma(const ConvArgs& a, int prec, hipStream_t st) {
  const int grid = ((a.M + 255) / 256) * ((a.ncols + 127) / 128);
  return prec == 1 ? launch_dma_prec<1>(a, grid, st) : prec == 4 ? launch_dma_prec<4>(a, grid, st) : launch_dma_prec<3>(a, grid, st);
}

// (A three-stage branch-free prefetch loop for tile_cfg 11 / 14 -- round 4 -- measured 0.5 ms per step slower than the two-stage loop
// on a live network and is kept as tools/probe/igemm_pipe3/igemm_pipe3.patch with its numbers.)
template <int BM, int BN, int PIPE>
int launch_cfg(const ConvArgs& a, int prec, hipStream_t st) {
  int mt = (a.M + BM - 1) / BM, nt = (a.ncols + BN - 1) / BN;
  dim3 grid(mt * nt), block(256);
  if (a.x_bf16) {   // bf16-stored input: plain-bf16 products on the two-deep-prefetch form of the tile
    if (prec != 1) return -7;
    constexpr bool k64 = true;   // 64-channel K steps on bf16-stored input (32.3 against 34.2 ms per 2-byte step with 32-channel steps)
    if (k64 && (a.cin_pad & 63) == 0 && (a.cin_valid & 7) == 0 && (a.ldx & 7) == 0)
      hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 1, 2, true, 64>), grid, block, 0, st, a);
    else
      hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 1, 2, true>), grid, block, 0, st, a);
  } else if (prec == 1)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 1, PIPE>), grid, block, 0, st, a);
  else if (prec == 0)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 0, PIPE>), grid, block, 0, st, a);   // exact fp32 (w_pk from zs3_prep_weight_f32)
  else if (prec == 4)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 4, PIPE>), grid, block, 0, st, a);   // fp16 hi/lo (w_pk prepared with f_fmt = 1)
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 3, PIPE>), grid, block, 0, st, a);
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
    int t = tile_cfg % 10;
    bm = (tile_cfg == 31 || tile_cfg == 41 || tile_cfg == 51 || tile_cfg == 141) ? 256
         : (tile_cfg == 42 || tile_cfg == 142) ? 192 : ((t == 3 || t == 4) ? 64 : 128);
  }
  return (M + bm - 1) / bm;
}

static int conv_igemm_impl(const float* x, const void* w_pk, float* y, const float* scale, const float* shift,
                           const float* res, float* stat_partial, int N, int H, int W, int Ho, int Wo, int cin_pad,
                           int cin_valid, int ldx, int KH, int KW, int stride, int pad_h, int pad_w, int dil, int ncols,
                           int ldy, int ldr, int act, float leak, int accumulate, int dgrad, int prec, int tile_cfg,
                           const void* zero_page, void* stream, const float* bs_y, int bs_ldy, const float* bs_mean,
                           const float* bs_istd, const float* bs_msc, const float* bs_msh,
                           const unsigned char* bs_mbits, float* bs_partial, const unsigned char* res_mbits, int io,
                           const float* in_scale = nullptr, const float* in_shift = nullptr) {
  if (cin_pad % 32 != 0 || cin_valid % 4 != 0 || ldx % 4 != 0 || (prec != 0 && prec != 1 && prec != 3 && prec != 4)) return -1;
  if (stride < 1 || (stride & (stride - 1)) != 0 || zero_page == nullptr) return -1;
  if (((uintptr_t)x & 15) || ((uintptr_t)w_pk & 15) || ((uintptr_t)zero_page & 15)) return -2;
  if (bs_partial && (!bs_y || !bs_mean || !bs_istd || (bs_ldy & 3) || (ncols & 3) || (ldy & 3) || (res && (ldr & 3))))
    return -6;   // the fused BN-backward sums need the vectorised store path
  if (res_mbits && (!res || (ncols & 3) || (ldy & 3) || (ldr & 3))) return -6;
  ConvArgs a;
  a.x = x; a.w_pk = (const unsigned short*)w_pk; a.y = y;
  a.scale = scale; a.shift = shift; a.res = res; a.stat_partial = stat_partial;
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.cin_pad = cin_pad; a.cin_valid = cin_valid; a.ldx = ldx;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.dil = dil;
  a.ncols = ncols; a.ldw = KH * KW * cin_pad; a.ldy = ldy; a.ldr = ldr; a.M = N * Ho * Wo;
  a.act = act; a.accumulate = accumulate; a.dgrad = dgrad; a.leak = leak;
  a.zero = (const float*)zero_page;
  a.bs_y = bs_y; a.bs_ldy = bs_ldy; a.bs_mean = bs_mean; a.bs_istd = bs_istd; a.bs_msc = bs_msc; a.bs_msh = bs_msh;
  a.bs_mbits = bs_mbits; a.bs_partial = bs_partial; a.res_mbits = res_mbits;
  a.x_bf16 = io & ZS3_IO_IN16 ? 1 : 0;
  a.y_bf16 = io & ZS3_IO_OUT16 ? 1 : 0;
  if (io & ~3) return -1;
  if (a.x_bf16 && (prec != 1 || in_scale)) return -7;   // bf16-stored input: plain-bf16 products, no producer-side transform
  a.in_scale = in_scale; a.in_shift = in_shift;
  if ((in_scale == nullptr) != (in_shift == nullptr) || ((uintptr_t)in_scale & 15) || ((uintptr_t)in_shift & 15)) return -1;
  a.stride_log2 = 0;
  while ((1 << a.stride_log2) < stride) ++a.stride_log2;
  if (a.M <= 0 || ncols <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int cfg = tile_cfg;
  if (cfg == 0) {
    int bn = ncols > 64 ? 128 : 64;
    long blocks = (long)((a.M + 127) / 128) * ((ncols + bn - 1) / bn);
    bool small = blocks < 512;
    cfg = bn == 128 ? (small ? 3 : 1) : (small ? 4 : 2);
  }
  if (in_scale && cfg != 41 && cfg != 42 && cfg != 51 && cfg != 52) return -7;   // only the producer-converting kernels transform x
  if (prec == 0 && cfg > 14) return -7;   // the exact-fp32 test mode exists on the register-staged kernel only
  if (a.x_bf16 && cfg == 31) return -7;   // the LDS-DMA kernel moves raw fp32 rows
  switch (cfg) {
    case 1: return launch_cfg<128, 128, 1>(a, prec, st);
    case 2: return launch_cfg<128, 64, 1>(a, prec, st);
    case 3: return launch_cfg<64, 128, 1>(a, prec, st);
    case 4: return launch_cfg<64, 64, 1>(a, prec, st);
    case 11: return launch_cfg<128, 128, 2>(a, prec, st);
    case 12: return launch_cfg<128, 64, 2>(a, prec, st);
    case 13: return launch_cfg<64, 128, 2>(a, prec, st);
    case 14: return launch_cfg<64, 64, 2>(a, prec, st);
    case 31: return launch_dma(a, prec, st);
    case 41: return zs3conv::launch_halo(a, 256, prec, st);   // -7: not a stride-1 same-size multi-tap layer (zs3_conv_halo_ok)
    case 42: return zs3conv::launch_halo(a, 192, prec, st);
    case 141: a.x_bf16 = 1; return zs3conv::launch_halo(a, 256, prec, st);   // (round-3 spelling of io bit 0 on tile_cfg 41 / 42)
    case 142: a.x_bf16 = 1; return zs3conv::launch_halo(a, 192, prec, st);
    case 51: return zs3conv::launch_pw(a, 256, prec, st);     // -7: not a 1x1 stride-1 layer (zs3_conv_pw_ok)
    case 52: return zs3conv::launch_pw(a, 128, prec, st);
  }
  return -3;
}

extern "C" int zs3_conv_igemm(const float* x, const void* w_pk, float* y, const float* scale,
                              const float* shift, const float* res, float* stat_partial, int N, int H, int W,
                              int Ho, int Wo, int cin_pad, int cin_valid, int ldx, int KH, int KW, int stride,
                              int pad_h, int pad_w, int dil, int ncols, int ldy, int ldr, int act, float leak,
                              int accumulate, int dgrad, int prec, int tile_cfg, const void* zero_page, int io, void* stream) {
  return conv_igemm_impl(x, w_pk, y, scale, shift, res, stat_partial, N, H, W, Ho, Wo, cin_pad, cin_valid, ldx, KH, KW,
                         stride, pad_h, pad_w, dil, ncols, ldy, ldr, act, leak, accumulate, dgrad, prec, tile_cfg,
                         zero_page, stream, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, io);
}

// zs3_conv_igemm with the input read through x' = max(x * in_scale[c] + in_shift[c], 0) (tile_cfg 41 / 42 / 51 / 52 only, -7 otherwise):
// the BatchNorm-apply + ReLU of the producing layer happens in the consumer's operand path and its activation is never stored.
extern "C" int zs3_conv_igemm_in(const float* x, const void* w_pk, float* y, const float* scale, const float* shift,
                                 const float* res, float* stat_partial, int N, int H, int W, int Ho, int Wo, int cin_pad,
                                 int cin_valid, int ldx, int KH, int KW, int stride, int pad_h, int pad_w, int dil, int ncols,
                                 int ldy, int ldr, int act, float leak, int accumulate, int dgrad, int prec, int tile_cfg,
                                 const void* zero_page, const float* in_scale, const float* in_shift, int io, void* stream) {
  if (!in_scale || !in_shift) return -1;
  return conv_igemm_impl(x, w_pk, y, scale, shift, res, stat_partial, N, H, W, Ho, Wo, cin_pad, cin_valid, ldx, KH, KW,
                         stride, pad_h, pad_w, dil, ncols, ldy, ldr, act, leak, accumulate, dgrad, prec, tile_cfg,
                         zero_page, stream, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, io, in_scale,
                         in_shift);
}

extern "C" int zs3_conv_igemm_bnstats(const float* x, const void* w_pk, float* y, const float* res,
                                      const unsigned char* res_mask_bits, int N, int H, int W, int Ho, int Wo, int cin_pad,
                                      int cin_valid, int ldx, int KH, int KW, int stride, int pad_h, int pad_w, int dil,
                                      int ncols, int ldy, int ldr, int accumulate, int dgrad, int prec, int tile_cfg,
                                      const void* zero_page, const float* bn_y, int bn_ldy, const float* bn_mean,
                                      const float* bn_invstd, const float* mask_scale, const float* mask_shift,
                                      const unsigned char* mask_bits, float* bn_partial, int io, void* stream) {
  if (!bn_partial && !res_mask_bits) return -1;
  return conv_igemm_impl(x, w_pk, y, nullptr, nullptr, res, nullptr, N, H, W, Ho, Wo, cin_pad, cin_valid, ldx, KH, KW,
                         stride, pad_h, pad_w, dil, ncols, ldy, ldr, 0, 0.f, accumulate, dgrad, prec, tile_cfg, zero_page,
                         stream, bn_y, bn_ldy, bn_mean, bn_invstd, mask_scale, mask_shift, mask_bits, bn_partial,
                         res_mask_bits, io);
}
