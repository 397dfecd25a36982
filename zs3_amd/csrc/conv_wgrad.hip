// Weight-gradient of a convolution as an MFMA GEMM over the pixel axis (gfx950 / MI355X).
//
//   dw[co, kh, kw, ci] = sum_m dy[m, co] * x[gather(m, kh, kw), ci]          m = (n, ho, wo)
//
// GEMM view per filter tap: C[co, ci] = dy^T x, reduction length = N*Ho*Wo pixels.  Both operands are
// NHWC fp32 with the *channel* axis contiguous while the MFMA fragments want 8 consecutive reduction
// indices per lane, i.e. a transpose.  It is done for free while staging: each thread loads the same
// 4 channels of two consecutive pixels, splits them into bf16 hi/lo, and packs the (pixel, pixel+1)
// pair of one channel into one 32-bit word; LDS holds [pixel-pair][channel] words, written with
// conflict-free ds_write_b128 and read back as 4 x ds_read_b32 per fragment (lanes = consecutive
// channels => conflict-free).  The reduction order inside an MFMA is a free permutation as long as
// A and B agree, which they do because both tiles use the same pair packing.
//
// Split-K over pixel chunks (blockIdx.y) gives parallelism for the small-Cout x Cin layers; partial
// [splitk][Cout][K] slabs are summed by zs3_wgrad_reduce in a fixed order (deterministic).
//
// Replaces convolution_backward(weight) for every nn.Conv2d / nn.Linear on the hot path (see conv_igemm.hip).
#include "common.h"
#include "zs3hip.h"
#include <stdlib.h>

namespace {

struct WgradArgs {
  const float* dy;   // (bf16 when the launch says so: `io` bit 0 = dy, bit 1 = x; strides in elements)
  const float* x;
  float* dw;
  const float* zero;   // >= 256 B of zeros for masked loads
  int N, H, W, Ho, Wo;
  int KH, KW, stride, pad_h, pad_w, dil;
  int co_read, co_write, ci_read, ci_write;
  int lddy, ldx, ldw, cin_w;
  int M, chunk;
  long slab;  // elements per split-K slab
  int fold;   // > 0: the KH taps of a KH x 1 filter over `fold`-channel rows are folded into the channel axis (see zs3_conv_wgrad)
};

// TDY / TX: element types of dy / x in memory (float or bf16_t; bf16 storage with plain-bf16 products only)
template <int BC, int BD, int PREC, typename TDY = float, typename TX = float>  // BC = dy-channel (co) tile, BD = x-channel (ci) tile
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
  const TDY* const dybase = reinterpret_cast<const TDY*>(p.dy);
  const TX* const xbase = reinterpret_cast<const TX*>(p.x);
  constexpr int PA = BC / 64, PB = BD / 64;   // pixel pairs per thread and stage
  constexpr int TM = BC / 64, TN = BD / 64;   // 32x32 tiles per wave
  constexpr int PLANE_A = 16 * BC, PLANE_B = 16 * BD;  // words per plane (16 pixel pairs)
  constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;
  __shared__ __attribute__((aligned(16))) unsigned smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_ci = (p.ci_write + BD - 1) / BD, tiles_co = (p.co_write + BC - 1) / BC;
  // 1-D grid, XCD-aware: every XCD (private L2) gets a contiguous range of (split, tile) pairs, so the tiles
  // that reduce over the same pixel chunk -- and therefore read the same dy / x rows -- share one L2
  const int ntiles = tiles_ci * tiles_co * p.KH * p.KW;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int split = b / ntiles;
  b -= split * ntiles;
  const int tci = b % tiles_ci; b /= tiles_ci;
  const int tco = b % tiles_co; b /= tiles_co;
  const int tap = b;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int co0 = tco * BC, ci0 = tci * BD;
  const int m_begin = split * p.chunk;
  const int m_end = min(p.M, m_begin + p.chunk);

  // staging coordinates: A (dy) -- BC/4 threads per pixel, B (x) -- BD/4 threads per pixel
  constexpr int TPA = BC / 4, TPB = BD / 4;
  const int cqa = (tid % TPA) * 4, pra = tid / TPA;  // pair index base for A: pra + (256/TPA)*i
  const int cqb = (tid % TPB) * 4, prb = tid / TPB;
  constexpr int SA = 256 / TPA, SB = 256 / TPB;       // pair stride between a thread's pairs
  const bool a_cok = (co0 + cqa) < p.co_read;
  const bool b_cok = (ci0 + cqb) < p.ci_read;

  f32x4 areg[PA][2], breg[PB][2];

  // Per-thread pixel slots.  A slot's pixel index advances by 32 every K step; (n, oh, ow) are carried
  // incrementally (one division when the block starts, none in the loop) and masked loads read the zero page,
  // so the loop body is branch-free apart from the wave-uniform trip count.
  const TDY* aptr[PA][2];
  int am[PA][2];
#pragma unroll
  for (int i = 0; i < PA; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      am[i][e] = m_begin + 2 * (pra + SA * i) + e;
      aptr[i][e] = dybase + (size_t)am[i][e] * p.lddy + co0 + cqa;
    }
  int bm[PB][2], bn[PB][2], boh[PB][2], bow[PB][2];
  {
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        int m = m_begin + 2 * (prb + SB * i) + e;
        bm[i][e] = m;
        int mm = m < p.M ? m : 0;
        int n = mm / hw, rem = mm - n * hw;
        bn[i][e] = n;
        boh[i][e] = rem / p.Wo;
        bow[i][e] = rem - boh[i][e] * p.Wo;
      }
  }
  // fold mode (the stem's 7x1 filter over 32-float windows): "input channel" c of this launch is (tap c / fold, channel c % fold),
  // so the tap is a property of the staging thread's channel quad, not of the workgroup: one workgroup covers several taps and
  // dy is read once per 128 folded channels instead of once per tap
  const int cfull = ci0 + cqb;
  const int kh_t = p.fold > 0 ? cfull / p.fold : kh;
  const int cch = p.fold > 0 ? cfull - kh_t * p.fold : cfull;
  const int tap_h = kh_t * p.dil - p.pad_h, tap_w = kw * p.dil - p.pad_w;

  auto load_tile = [&]() {
#pragma unroll
    for (int i = 0; i < PA; ++i)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const TDY* ptr = (a_cok && am[i][e] < m_end) ? aptr[i][e] : reinterpret_cast<const TDY*>(p.zero);
        areg[i][e] = ld4<TDY>(ptr);
        am[i][e] += 32;
        aptr[i][e] += (size_t)32 * p.lddy;
      }
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int hi = boh[i][e] * p.stride + tap_h, wi = bow[i][e] * p.stride + tap_w;
        const bool ok = b_cok && bm[i][e] < m_end && ((hi | wi) >= 0) && hi < p.H && wi < p.W;
        const TX* ptr = ok ? xbase + ((((size_t)bn[i][e] * p.H + hi) * p.W + wi) * p.ldx + cch) : reinterpret_cast<const TX*>(p.zero);
        breg[i][e] = ld4<TX>(ptr);
        // advance this slot by 32 pixels
        bm[i][e] += 32;
        bow[i][e] += 32;
        while (bow[i][e] >= p.Wo) {
          bow[i][e] -= p.Wo;
          if (++boh[i][e] == p.Ho) {
            boh[i][e] = 0;
            ++bn[i][e];
          }
        }
      }
  };
  auto store_tile = [&](int stage) {
    unsigned* As = smem + stage * STAGE;
    unsigned* Bs = As + 2 * PLANE_A;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      u32x4 hi, lo;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned h, l;
        if constexpr (PREC == 0) {   // exact fp32 (test mode): plane 0 = the even pixel of the pair, plane 1 = the odd one, raw
          h = __float_as_uint(areg[i][0][c]);
          l = __float_as_uint(areg[i][1][c]);
        } else {
          split_pair<PREC>(areg[i][0][c], areg[i][1][c], h, l);
        }
        hi[c] = h;
        lo[c] = l;
      }
      unsigned* dst = As + (pra + SA * i) * BC + cqa;
      *reinterpret_cast<u32x4*>(dst) = hi;
      if (PREC != 1) *reinterpret_cast<u32x4*>(dst + PLANE_A) = lo;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      u32x4 hi, lo;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned h, l;
        if constexpr (PREC == 0) {
          h = __float_as_uint(breg[i][0][c]);
          l = __float_as_uint(breg[i][1][c]);
        } else {
          split_pair<PREC>(breg[i][0][c], breg[i][1][c], h, l);
        }
        hi[c] = h;
        lo[c] = l;
      }
      unsigned* dst = Bs + (prb + SB * i) * BD + cqb;
      *reinterpret_cast<u32x4*>(dst) = hi;
      if (PREC != 1) *reinterpret_cast<u32x4*>(dst + PLANE_B) = lo;
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
    if constexpr (PREC == 0) {
      // v_mfma_f32_32x32x2_f32 per pixel pair t: lanes 0-31 supply the even pixel (plane 0), lanes 32-63 the odd one (plane 1)
      const unsigned* A0 = smem + stage * STAGE + (lane >> 5) * PLANE_A + wm * (BC / 2) + (lane & 31);
      const unsigned* B0 = smem + stage * STAGE + 2 * PLANE_A + (lane >> 5) * PLANE_B + wn * (BD / 2) + (lane & 31);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = __uint_as_float(A0[t * BC + i * 32]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = __uint_as_float(B0[t * BD + j * 32]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      return;
    }
    const unsigned* As = smem + stage * STAGE + wm * (BC / 2) + (lane & 31);
    const unsigned* Bs = smem + stage * STAGE + 2 * PLANE_A + wn * (BD / 2) + (lane & 31);
    const int prow = (lane >> 5) * 4;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      union Frag { bf16x8 v; unsigned u[4]; };
      Frag a_hi[TM], a_lo[TM], b_hi[TN], b_lo[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a_hi[i].u[q] = As[(kk * 8 + prow + q) * BC + i * 32];
          if (PREC == 3) a_lo[i].u[q] = As[PLANE_A + (kk * 8 + prow + q) * BC + i * 32];
        }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          b_hi[j].u[q] = Bs[(kk * 8 + prow + q) * BD + j * 32];
          if (PREC == 3) b_lo[j].u[q] = Bs[PLANE_B + (kk * 8 + prow + q) * BD + j * 32];
        }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (PREC == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[i].v, b_hi[j].v, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i].v, b_lo[j].v, acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i].v, b_hi[j].v, acc[i][j], 0, 0, 0);
        }
    }
  };

  const int KT = (m_end - m_begin + 31) / 32;
  if (KT > 0) {
    load_tile();
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int cur = kt & 1;
      const bool more = kt + 1 < KT;
      if (more) load_tile();
      compute(cur);
      if (more) store_tile(cur ^ 1);
      __syncthreads();
    }
  }

  float* out = p.dw + (size_t)split * p.slab;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ci = ci0 + wn * (BD / 2) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * (BC / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.co_write && ci < p.ci_write) out[(size_t)co * p.ldw + (size_t)tap * p.cin_w + ci] = acc[i][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant for layers with >= 256 channels on both sides (kernel 2 of zs3_conv_wgrad).
//
// Block = 256(co) x 256(ci) outputs of one filter tap and one pixel chunk; 8 waves (2 per SIMD) with 128(co) x
// 64(ci) wave tiles.  Per K step (16 pixels) the block needs 16 dy rows and 16 gathered x rows: one
// global_load_lds_dwordx4 per pixel row (64 lanes x 16 B = the 256 fp32 channels of that pixel) moves them straight
// into a 4-slot LDS ring (32 KB per slot), three tiles ahead of the multiply, waiting only on a counted vmcnt;
// each wave issues 2 + 2 rows.  LDS therefore holds raw fp32 [pixel][channel]; the transpose the MFMA wants (8
// consecutive pixels of one channel per lane) is done with ds_read_b32 (lanes = consecutive channels:
// conflict-free) and the bf16 hi/lo split happens in registers.  24 MFMAs per K step and wave; the LDS reads,
// conversion VALU ops and DMA issues that prepare the following fragments / tiles are pinned into the slots between
// them (6 VALU per MFMA -- more than one wave can hide, which is what the second wave on the SIMD is for).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

template <int PREC>
__global__ __launch_bounds__(512) void conv_wgrad_dma_kernel(const WgradArgs p) {
  constexpr int BC = 256, BD = 256, KPIX = 16;
  constexpr int PART = KPIX * 1024, STAGE_BYTES = 2 * PART;   // dy rows then x rows, 1 KB (256 fp32) per pixel
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_ci = (p.ci_write + BD - 1) / BD, tiles_co = (p.co_write + BC - 1) / BC;
  const int ntiles = tiles_ci * tiles_co * p.KH * p.KW;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int split = b / ntiles;
  b -= split * ntiles;
  const int tci = b % tiles_ci; b /= tiles_ci;
  const int tco = b % tiles_co; b /= tiles_co;
  const int tap = b;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int co0 = tco * BC, ci0 = tci * BD;
  const int m_begin = split * p.chunk;
  const int m_end = min(p.M, m_begin + p.chunk);
  const int KT = (m_end - m_begin + KPIX - 1) / KPIX;

  // ---- loader state: this wave fetches rows 2 wave, 2 wave + 1 of both parts of every tile
  const bool a_cok = co0 + lane * 4 < p.co_read;   // this lane's 4 channels exist
  const bool b_cok = ci0 + lane * 4 < p.ci_read;
  const int tap_h = kh * p.dil - p.pad_h, tap_w = kw * p.dil - p.pad_w;
  int pm[2], pn[2], poh[2], pow_[2];   // wave-uniform pixel slots, advanced by 16 pixels per tile without divisions
  {
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      int m = m_begin + 2 * wave + r;
      pm[r] = m;
      int mm = m < p.M ? m : 0;
      pn[r] = mm / hw;
      int rem = mm - pn[r] * hw;
      poh[r] = rem / p.Wo;
      pow_[r] = rem - poh[r] * p.Wo;
    }
  }
  // branch-free (the K loop must stay one basic block for the pinned schedule): wave-uniform row bases, per-lane
  // select against the zero page, and single-wrap carries (the launcher guarantees Wo >= KPIX)
  const unsigned long zaddr = (unsigned long)p.zero;
  const unsigned lane16 = lane * 16;
  const long lane_a = a_cok ? -1L : 0L, lane_b = b_cok ? -1L : 0L;
  auto issue_row = [&](int slot, int r) {   // masks instead of conditions: hipcc turns wave-uniform conditions into branches
    unsigned char* sA = dsm + slot * STAGE_BYTES + wave * 2048 + r * 1024;
    const int in = (pm[r] - m_end) >> 31;   // all ones while the pixel is inside this split's chunk
    const unsigned long ga = (unsigned long)(p.dy + (size_t)pm[r] * p.lddy + co0) + lane16;
    const unsigned long pa = zaddr + ((ga - zaddr) & (unsigned long)((long)in & lane_a));
    __builtin_amdgcn_global_load_lds((gbl_void_t*)pa, (lds_void_t*)sA, 16, 0, 0);
    const int hi = poh[r] * p.stride + tap_h, wi = pow_[r] * p.stride + tap_w;
    const int rok = in & ~((hi | wi) >> 31) & ((hi - p.H) >> 31) & ((wi - p.W) >> 31);
    const int hic = hi & rok, wic = wi & rok;
    const unsigned long gb = (unsigned long)(p.x + (((size_t)pn[r] * p.H + hic) * p.W + wic) * p.ldx + ci0) + lane16;
    const unsigned long pb = zaddr + ((gb - zaddr) & (unsigned long)((long)rok & lane_b));
    __builtin_amdgcn_global_load_lds((gbl_void_t*)pb, (lds_void_t*)(sA + PART), 16, 0, 0);
    pm[r] += KPIX;
    pow_[r] += KPIX;
    const int cw = (p.Wo - 1 - pow_[r]) >> 31;   // all ones when the column wrapped (Wo >= KPIX: at most once)
    pow_[r] -= p.Wo & cw;
    poh[r] -= cw;
    const int ch = (p.Ho - 1 - poh[r]) >> 31;
    poh[r] -= p.Ho & ch;
    pn[r] -= ch;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int wco = wave >> 2, wci = wave & 3;
  const int c = lane & 31, h = lane >> 5;
  // byte offset of this lane's first value inside a slot: pixel row 8h, channel (wave base + c)
  const int offA = (8 * h) * 1024 + (128 * wco + c) * 4;
  const int offB = PART + (8 * h) * 1024 + (64 * wci + c) * 4;
  u32x4 a_hi[2], a_lo[2], b_hi[2][2], b_lo[2][2];
  float raw1[8], raw2[8];
  float t_ha = 0.f, t_hb = 0.f;
  auto read_frag = [&](float (&raw)[8], const unsigned char* base) {
#pragma unroll
    for (int e = 0; e < 8; ++e) raw[e] = *reinterpret_cast<const float*>(base + e * 1024);
  };
  auto split_half = [&](const float (&raw)[8], u32x4& hi, u32x4& lo, int hp) {   // hp = 2*q + phase
    const int q = hp >> 1;
    if ((hp & 1) == 0) {
      const unsigned hw = cvt_pk_bf16(raw[2 * q], raw[2 * q + 1]);
      hi[q] = hw;
      t_ha = __uint_as_float(hw << 16);
      t_hb = __uint_as_float(hw & 0xFFFF0000u);
    } else {
      lo[q] = PREC == 3 ? cvt_pk_bf16(raw[2 * q] - t_ha, raw[2 * q + 1] - t_hb) : 0u;
    }
  };

  // tiles 0..2 in flight (4 loads per tile and wave); tiles 0 and 1 must have landed before the first K step
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 2; ++r) issue_row(t, r);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {   // prologue: A fragment 0 and the two B fragments of tile 0
    read_frag(raw1, dsm + offA);
#pragma unroll
    for (int hp = 0; hp < 8; ++hp) split_half(raw1, a_hi[0], a_lo[0], hp);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      read_frag(raw2, dsm + offB + j * 128);
#pragma unroll
      for (int hp = 0; hp < 8; ++hp) split_half(raw2, b_hi[0][j], b_lo[0][j], hp);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // kstep: 4 groups of 6 MFMAs (A fragment i against the 2 B fragments of the current tile).  The slots in between
  // read and split A fragment i+1 (i == 3: fragment 0 of the next tile) and half of a B fragment of the next tile
  // (groups 0,1: fragment 0; groups 2,3: fragment 1), and issue this wave's rows of the tile three steps ahead.
  auto kstep = [&](const unsigned char* cur, const unsigned char* nxt, int bcur, int slot3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int abuf = i & 1, bj = i >> 1, bhalf = i & 1;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, a_hi[abuf]);
      const bf16x8 al = __builtin_bit_cast(bf16x8, a_lo[abuf]);
#pragma unroll
      for (int g = (PREC == 3 ? 0 : 4); g < 6; ++g) {
        const int t = g >> 1, j = g & 1;
        const bf16x8 bb = __builtin_bit_cast(bf16x8, t == 1 ? b_lo[bcur][j] : b_hi[bcur][j]);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t == 0 ? al : ah, bb, acc[i][j], 0, 0, 0);
        const int f = PREC == 3 ? g : g - 4;   // filler slot index (PREC 1 has only two MFMAs per group)
        if (f == 0) {
          read_frag(raw1, (i < 3 ? cur + (i + 1) * 128 : nxt) + offA);
          if (bhalf == 0) read_frag(raw2, nxt + offB + bj * 128);
          if (i < 2) issue_row(slot3, i);   // past the last tile these are zero-page loads into a free slot
        }
        if (PREC == 3) {
          if (g >= 2) {   // A pairs 0..3 in slots 2..5
            split_half(raw1, a_hi[abuf ^ 1], a_lo[abuf ^ 1], 2 * (g - 2));
            split_half(raw1, a_hi[abuf ^ 1], a_lo[abuf ^ 1], 2 * (g - 2) + 1);
          }
          if (g >= 4) {   // two of the four B pairs per group, in slots 4,5
            const int q = 2 * bhalf + (g - 4);
            split_half(raw2, b_hi[bcur ^ 1][bj], b_lo[bcur ^ 1][bj], 2 * q);
            split_half(raw2, b_hi[bcur ^ 1][bj], b_lo[bcur ^ 1][bj], 2 * q + 1);
          }
        } else if (f == 1) {
#pragma unroll
          for (int hp = 0; hp < 8; ++hp) split_half(raw1, a_hi[abuf ^ 1], a_lo[abuf ^ 1], hp);
#pragma unroll
          for (int q = 2 * bhalf; q < 2 * bhalf + 2; ++q) {
            split_half(raw2, b_hi[bcur ^ 1][bj], b_lo[bcur ^ 1][bj], 2 * q);
            split_half(raw2, b_hi[bcur ^ 1][bj], b_lo[bcur ^ 1][bj], 2 * q + 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // tile kt+2 must have landed before anybody starts step kt+1 (which reads it); tile kt+3 may stay in flight
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  int st = 0;
  for (int kt = 0; kt < KT; kt += 2) {   // two K steps per trip: the B double-buffer index stays a compile-time constant
    const int st1 = (st + 1) & 3, st2 = (st + 2) & 3, st3 = (st + 3) & 3;
    kstep(dsm + st * STAGE_BYTES, dsm + st1 * STAGE_BYTES, 0, st3);
    if (kt + 1 < KT) kstep(dsm + st1 * STAGE_BYTES, dsm + st2 * STAGE_BYTES, 1, st);
    st = st2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (masked) row loads must land before the LDS is released

  float* out = p.dw + (size_t)split * p.slab;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ci = ci0 + wci * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.co_write && ci < p.ci_write) out[(size_t)co * p.ldw + (size_t)tap * p.cin_w + ci] = acc[i][j][r];
      }
    }
}

template <int PREC>
int launch_wgrad_dma_prec(const WgradArgs& a, int taps, int splitk, hipStream_t st) {
  constexpr int LDS_BYTES = 4 * 2 * 16 * 1024;   // 128 KB ring
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_kernel<PREC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return -4;
    configured = true;
  }
  int tiles = ((a.co_write + 255) / 256) * ((a.ci_write + 255) / 256) * taps;
  hipLaunchKernelGGL((conv_wgrad_dma_kernel<PREC>), dim3(tiles * splitk), dim3(512), LDS_BYTES, st, a);
  return ZS3_LAUNCH_CHECK();
}

// dw = sum over the split-K slabs (fixed order: deterministic); n4 = slab / 4 when slab and the pointers allow 16-byte accesses
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long n, int splitk, long slab) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splitk; ++k) s += part[(size_t)k * slab + i];
    dw[i] = s;
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const f32x4* __restrict__ part, f32x4* __restrict__ dw, long n4,
                                                           int splitk, long slab4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 s = part[i];
#pragma unroll 4
    for (int k = 1; k < splitk; ++k) s += part[(size_t)k * slab4 + i];
    dw[i] = s;
  }
}

// many slabs (the stem: hundreds of splits of a 57 KB gradient): 64 columns x 16 slab groups per block, thread (tx, ty) sums the
// slabs ty, ty + 16, ... of its column, the groups are combined through LDS in a fixed order (deterministic) -- a thread of the
// plain form above would walk every slab in one dependent chain
constexpr int RED_GROUPS = 16;
__global__ __launch_bounds__(64 * RED_GROUPS) void wgrad_reduce4_tall_kernel(const f32x4* __restrict__ part, f32x4* __restrict__ dw,
                                                                           long n4, int splitk, long slab4) {
  __shared__ f32x4 red[RED_GROUPS][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + tx;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
#pragma unroll 4
    for (int k = ty; k < splitk; k += RED_GROUPS) s += part[(size_t)k * slab4 + i];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < n4) {
#pragma unroll
    for (int g = 1; g < RED_GROUPS; ++g) s += red[g][tx];
    dw[i] = s;
  }
}

template <int BC, int BD>
int launch_wgrad(const WgradArgs& a, int taps, int splitk, int prec, int io, hipStream_t st) {
  int tiles = ((a.co_write + BC - 1) / BC) * ((a.ci_write + BD - 1) / BD) * taps;
  dim3 grid(tiles * splitk), block(256);
  if (io) {   // bf16-stored dy (bit 0) and / or x (bit 1): plain-bf16 products
    if (prec != 1) return -7;
    if (io == 3) hipLaunchKernelGGL((conv_wgrad_kernel<BC, BD, 1, bf16_t, bf16_t>), grid, block, 0, st, a);
    else if (io == 1) hipLaunchKernelGGL((conv_wgrad_kernel<BC, BD, 1, bf16_t, float>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<BC, BD, 1, float, bf16_t>), grid, block, 0, st, a);
  } else if (prec == 1)
    hipLaunchKernelGGL((conv_wgrad_kernel<BC, BD, 1>), grid, block, 0, st, a);
  else if (prec == 0)
    hipLaunchKernelGGL((conv_wgrad_kernel<BC, BD, 0>), grid, block, 0, st, a);   // exact fp32 (test mode)
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<BC, BD, 3>), grid, block, 0, st, a);
  return ZS3_LAUNCH_CHECK();
}

// CUs one launch is sized for (split-K factors and grids aim at this many workgroup slots), ZS3_WGRAD_CUS.  The launches run on
// two side streams next to the dgrad chain; same-box sweep of the supervised step (tools/probe/r2r.sh, ms per step): 64 -> 54.4,
// 96 -> 48.3, 128 -> 48.9, 176 -> 48.4, 256 on one stream -> 49.8, 96 on three streams -> 49.1.
static int wgrad_cus() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("ZS3_WGRAD_CUS");
    v = e ? atoi(e) : 96;
    if (v < 8 || v > 256) v = 96;
  }
  return v;
}

int pick_splitk(int M, int tiles) {
  int chunks = (M + 31) / 32;
  int want = (3 * wgrad_cus() + tiles - 1) / tiles;   // aim at >= ~768 workgroups (3 per CU); every split costs a [Cout][K] slab of traffic
  if (want < 1) want = 1;
  int maxs = chunks / 16;                 // at least 16 K-steps (512 pixels) per split
  if (maxs < 1) maxs = 1;
  int s = want < maxs ? want : maxs;
  if (s > 256) s = 256;
  return s;
}

}  // namespace

// 128-wide tiles whenever there are more than 64 channels (measured: a half-empty 128 tile still beats 64-wide tiles)
static int pick_tile_dim(int c) { return c > 64 ? 128 : 64; }
static int tile_override() { return 0; }   // (64: force 64 x 64 tiles; a debug knob of round 1)
// kernel 2 (LDS-DMA, 256x256 tiles) takes the leading multiple-of-256 input channels when Cout fills 256-wide tiles;
// a remainder of <= 128 input channels (the 304-channel decoder concat) goes to kernel 1 in a second launch over the
// same split-K slabs.  Returns the number of input channels given to kernel 2.  zs3_conv_wgrad_set_kernel(1 | 2) forces one.
static int g_wgrad_kernel = -1;   // zs3_conv_wgrad_set_kernel (-1: not set yet = 0, the rules)
static int dma_width(int co, int ci, int wo, int M) {
  if (g_wgrad_kernel < 0) g_wgrad_kernel = 0;
  const int v = g_wgrad_kernel;
  if (v == 1) return 0;
  if (v == 2) return ci;
  // short reductions (the GMMN generator's per-class pixel sets) stay on kernel 1: the 256x256 blocks' prologue, tile store
  // and the second launch for a ragged Cin cost more than they save there (measured: +7 ms per GMMN step)
  if (co < 256 || co % 256 || ci < 256 || wo < 16 || M < 8192) return 0;
  const int rem = ci % 256;
  return rem <= 128 ? ci - rem : 0;
}
static int pick_splitk_dma(int M, int tiles, long out_elems) {
  // One 512-thread block per CU, 256 CUs.  Cost model in units of one K step (16 pixels, ~1.5 us): rounds x (K steps
  // per block + ~12 for prologue and the 256 KB tile store) + the [Cout][K] slab each split writes and the reduction
  // reads back (priced at ~1 TB/s: in the training step this kernel shares HBM with the BN / dgrad stream; measured).  A 257th block costs a whole extra round, so the block count matters more than the split.
  int maxs = M / 256;   // at least 16 K-steps (256 pixels) per split
  if (maxs < 1) maxs = 1;
  if (maxs > 128) maxs = 128;
  const int cus = wgrad_cus();
  const double slab_units = (double)out_elems * 4.0 * 2.0 / 1e12 / 1.5e-6;
  const long ksteps = (M + 15) / 16;
  int best = 1;
  double best_t = 1e30;
  for (int s = 1; s <= maxs; ++s) {
    const long n = (long)tiles * s;
    const double t = (double)((n + cus - 1) / cus) * ((double)((ksteps + s - 1) / s) + 12.0) + (s > 1 ? s * slab_units : 0.0);
    if (t < best_t) {
      best_t = t;
      best = s;
    }
  }
  return best;
}
// 0: the rules above, 1: register-staged kernel for every layer (what the exact-fp32 test mode, prec = 0, needs), 2: LDS-DMA kernel
// wherever Cin allows; returns the previous value.  Plans (zs3_conv_wgrad_plan) made under another setting are stale.
extern "C" int zs3_conv_wgrad_set_kernel(int kernel) {
  if (g_wgrad_kernel < 0) g_wgrad_kernel = 0;
  const int old = g_wgrad_kernel;
  if (kernel >= 0 && kernel <= 2) g_wgrad_kernel = kernel;
  return old;
}
#ifndef ZS3_STEM_WGS
// same-box A/B of the supervised step (tools/probe/r5t.sh; the side streams' sizing gave 288 workgroups = 144 splits): 288 -> 45.05 /
// 45.13 ms, 512 -> 44.75 / 44.74, 768 -> 44.64 / 44.62, 1536 -> 44.73 / 44.69, 3072 -> 44.70 / 44.72; the launch itself 361 -> 190 us
#define ZS3_STEM_WGS 768
#endif
constexpr int STEM_FOLD = 32;   // channel count of the stem's NHWC4 windows (8 pixels x 4 channels)
extern "C" int zs3_conv_wgrad_plan(int M, int Wo, int co, int ci, int taps, int* splitk_out, long* workspace_floats) {
  int s;
  const int wd = dma_width(co, ci, Wo, M);
  if (wd > 0) {
    s = pick_splitk_dma(M, ((co + 255) / 256) * ((wd + 255) / 256) * taps, (long)co * ci * taps);
  } else {
    int bc = pick_tile_dim(co), bd = pick_tile_dim(ci);
    if (tile_override() == 64) { bc = 64; bd = 64; }
    int tiles = ((co + bc - 1) / bc) * ((ci + bd - 1) / bd) * taps;
    if (ci == STEM_FOLD && taps == 7) {   // the folded stem launch of zs3_conv_wgrad: 224 channels, one tap
      bd = pick_tile_dim(ci * taps);
      tiles = ((co + bc - 1) / bc) * ((ci * taps + bd - 1) / bd);
      // the last launch of backward, on the main stream with nothing left to share the chip with: sized for all of it
      // (ZS3_STEM_WGS workgroups), not for the side streams' share
      s = (ZS3_STEM_WGS + tiles - 1) / tiles;
      const int maxs = (M + 31) / 32 / 16;
      if (s > maxs) s = maxs;
      if (s < 1) s = 1;
    } else
      s = pick_splitk(M, tiles);
  }
  *splitk_out = s;
  *workspace_floats = s > 1 ? (long)s * co * taps * ci : 0;
  return 0;
}

extern "C" int zs3_conv_wgrad(const float* dy, const float* x, float* dw, float* workspace, int N, int H, int W, int Ho,
                              int Wo, int KH, int KW, int stride, int pad_h, int pad_w, int dil, int co_read,
                              int co_write, int ci_read, int ci_write, int lddy, int ldx, int prec,
                              const void* zero_page, int io, void* stream) {
  if (io & ~3) return -1;
  if (co_read % 4 || ci_read % 4 || lddy % 4 || ldx % 4 || (prec != 0 && prec != 1 && prec != 3) || zero_page == nullptr) return -1;
  if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)zero_page & 15)) return -2;
  if ((prec == 0 || io) && dma_width(co_write, ci_write, Wo, N * Ho * Wo) > 0) return -7;   // exact fp32 / bf16 storage: zs3_conv_wgrad_set_kernel(1) first
  WgradArgs a;
  a.dy = dy; a.x = x; a.zero = (const float*)zero_page;
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.dil = dil;
  a.co_read = co_read; a.co_write = co_write; a.ci_read = ci_read; a.ci_write = ci_write;
  a.lddy = lddy; a.ldx = ldx; a.cin_w = ci_write; a.ldw = KH * KW * ci_write; a.fold = 0;
  a.M = N * Ho * Wo;
  const int taps = KH * KW;
  int splitk;
  long ws;
  zs3_conv_wgrad_plan(a.M, Wo, co_write, ci_write, taps, &splitk, &ws);
  const int wd = dma_width(co_write, ci_write, Wo, a.M);
  if (splitk > 1 && workspace == nullptr) return -3;
  int chunks = (a.M + 31) / 32;
  a.chunk = ((chunks + splitk - 1) / splitk) * 32;
  a.slab = (long)co_write * a.ldw;
  a.dw = splitk > 1 ? workspace : dw;
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  if (wd > 0) {   // leading wd input channels: LDS-DMA kernel
    WgradArgs d = a;
    d.ci_write = wd;
    d.ci_read = ci_read < wd ? ci_read : wd;
    rc = prec == 1 ? launch_wgrad_dma_prec<1>(d, taps, splitk, st) : launch_wgrad_dma_prec<3>(d, taps, splitk, st);
    if (rc) return rc;
  }
  if (wd < ci_write) {   // remaining (or all) input channels: register-staged kernel, same slabs and pixel chunks
    WgradArgs r = a;
    r.x = a.x + wd;      // (wd > 0 only with fp32 storage)
    r.dw = a.dw + wd;
    r.ci_write = ci_write - wd;
    r.ci_read = ci_read - wd;
    int taps_r = taps;
    if (wd == 0 && KH == 7 && KW == 1 && pad_w == 0 && ci_write == STEM_FOLD && ci_read == STEM_FOLD) {
      // the stem (7x7/s2 as a 7x1 filter over 32-float windows, resnet.py:73): per tap the tile would be 64 x 32 of a 64 x 64
      // MFMA tile and the 271 MB gradient would be read seven times (574 us at B=16).  [co][kh][1][32] weights are [co][224]
      // rows, so the taps fold into the channel axis: one tap, 224 channels, the staging thread derives its tap from its channel.
      r.fold = STEM_FOLD;
      r.KH = 1;
      r.ci_write = r.ci_read = KH * STEM_FOLD;
      r.cin_w = KH * STEM_FOLD;
      taps_r = 1;
    }
    int bc = pick_tile_dim(r.co_write), bd = pick_tile_dim(r.ci_write);
    if (tile_override() == 64) { bc = 64; bd = 64; }
    if (bc == 128 && bd == 128) rc = launch_wgrad<128, 128>(r, taps_r, splitk, prec, io, st);
    else if (bc == 128) rc = launch_wgrad<128, 64>(r, taps_r, splitk, prec, io, st);
    else if (bd == 128) rc = launch_wgrad<64, 128>(r, taps_r, splitk, prec, io, st);
    else rc = launch_wgrad<64, 64>(r, taps_r, splitk, prec, io, st);
  }
  if (rc) return rc;
  if (splitk > 1) {
    long n = a.slab;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if ((n & 3) == 0 && (((uintptr_t)workspace | (uintptr_t)dw) & 15) == 0) {
      blocks = (int)((n / 4 + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      if (splitk >= 4 * RED_GROUPS)
        hipLaunchKernelGGL(wgrad_reduce4_tall_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(64 * RED_GROUPS), 0, st,
                           (const f32x4*)workspace, (f32x4*)dw, n / 4, splitk, a.slab / 4);
      else
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(blocks), dim3(256), 0, st, (const f32x4*)workspace, (f32x4*)dw, n / 4,
                           splitk, a.slab / 4);
    } else {
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, n, splitk, a.slab);
    }
    rc = ZS3_LAUNCH_CHECK();
  }
  return rc;
}
