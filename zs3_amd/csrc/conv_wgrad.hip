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
  const float* dy;
  const float* x;
  float* dw;
  const float* zero;   // >= 256 B of zeros for masked loads
  int N, H, W, Ho, Wo;
  int KH, KW, stride, pad_h, pad_w, dil;
  int co_read, co_write, ci_read, ci_write;
  int lddy, ldx, ldw, cin_w;
  int M, chunk;
  long slab;  // elements per split-K slab
};

template <int BC, int BD, int PREC>  // BC = dy-channel (co) tile, BD = x-channel (ci) tile
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
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
  const float* aptr[PA][2];
  int am[PA][2];
#pragma unroll
  for (int i = 0; i < PA; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      am[i][e] = m_begin + 2 * (pra + SA * i) + e;
      aptr[i][e] = p.dy + (size_t)am[i][e] * p.lddy + co0 + cqa;
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
  const int tap_h = kh * p.dil - p.pad_h, tap_w = kw * p.dil - p.pad_w;

  auto load_tile = [&]() {
#pragma unroll
    for (int i = 0; i < PA; ++i)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float* ptr = (a_cok && am[i][e] < m_end) ? aptr[i][e] : p.zero;
        areg[i][e] = *reinterpret_cast<const f32x4*>(ptr);
        am[i][e] += 32;
        aptr[i][e] += (size_t)32 * p.lddy;
      }
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int hi = boh[i][e] * p.stride + tap_h, wi = bow[i][e] * p.stride + tap_w;
        const bool ok = b_cok && bm[i][e] < m_end && ((hi | wi) >= 0) && hi < p.H && wi < p.W;
        const float* ptr = ok ? p.x + ((((size_t)bn[i][e] * p.H + hi) * p.W + wi) * p.ldx + ci0 + cqb) : p.zero;
        breg[i][e] = *reinterpret_cast<const f32x4*>(ptr);
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
        split_pair<PREC>(areg[i][0][c], areg[i][1][c], h, l);
        hi[c] = h;
        lo[c] = l;
      }
      unsigned* dst = As + (pra + SA * i) * BC + cqa;
      *reinterpret_cast<u32x4*>(dst) = hi;
      if (PREC == 3) *reinterpret_cast<u32x4*>(dst + PLANE_A) = lo;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      u32x4 hi, lo;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned h, l;
        split_pair<PREC>(breg[i][0][c], breg[i][1][c], h, l);
        hi[c] = h;
        lo[c] = l;
      }
      unsigned* dst = Bs + (prb + SB * i) * BD + cqb;
      *reinterpret_cast<u32x4*>(dst) = hi;
      if (PREC == 3) *reinterpret_cast<u32x4*>(dst + PLANE_B) = lo;
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

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long n, int splitk, long slab) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splitk; ++k) s += part[(size_t)k * slab + i];
    dw[i] = s;
  }
}

template <int BC, int BD>
int launch_wgrad(const WgradArgs& a, int taps, int splitk, int prec, hipStream_t st) {
  int tiles = ((a.co_write + BC - 1) / BC) * ((a.ci_write + BD - 1) / BD) * taps;
  dim3 grid(tiles * splitk), block(256);
  if (prec == 1)
    hipLaunchKernelGGL((conv_wgrad_kernel<BC, BD, 1>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<BC, BD, 3>), grid, block, 0, st, a);
  return ZS3_LAUNCH_CHECK();
}

int pick_splitk(int M, int tiles) {
  int chunks = (M + 31) / 32;
  int want = (768 + tiles - 1) / tiles;   // aim at >= ~768 workgroups (3 per CU); every split costs a [Cout][K] slab of traffic
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
static int tile_override() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("ZS3_WGRAD_TILE"); v = e ? atoi(e) : 0; }
  return v;
}
extern "C" int zs3_conv_wgrad_plan(int M, int co, int ci, int taps, int* splitk_out, long* workspace_floats) {
  int bc = pick_tile_dim(co), bd = pick_tile_dim(ci);
  if (tile_override() == 64) { bc = 64; bd = 64; }
  int tiles = ((co + bc - 1) / bc) * ((ci + bd - 1) / bd) * taps;
  int s = pick_splitk(M, tiles);
  *splitk_out = s;
  *workspace_floats = s > 1 ? (long)s * co * taps * ci : 0;
  return 0;
}

extern "C" int zs3_conv_wgrad(const float* dy, const float* x, float* dw, float* workspace, int N, int H, int W, int Ho,
                              int Wo, int KH, int KW, int stride, int pad_h, int pad_w, int dil, int co_read,
                              int co_write, int ci_read, int ci_write, int lddy, int ldx, int prec,
                              const void* zero_page, void* stream) {
  if (co_read % 4 || ci_read % 4 || lddy % 4 || ldx % 4 || (prec != 1 && prec != 3) || zero_page == nullptr) return -1;
  if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)zero_page & 15)) return -2;
  WgradArgs a;
  a.dy = dy; a.x = x; a.zero = (const float*)zero_page;
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.dil = dil;
  a.co_read = co_read; a.co_write = co_write; a.ci_read = ci_read; a.ci_write = ci_write;
  a.lddy = lddy; a.ldx = ldx; a.cin_w = ci_write; a.ldw = KH * KW * ci_write;
  a.M = N * Ho * Wo;
  const int taps = KH * KW;
  int splitk;
  long ws;
  zs3_conv_wgrad_plan(a.M, co_write, ci_write, taps, &splitk, &ws);
  if (splitk > 1 && workspace == nullptr) return -3;
  int chunks = (a.M + 31) / 32;
  a.chunk = ((chunks + splitk - 1) / splitk) * 32;
  a.slab = (long)co_write * a.ldw;
  a.dw = splitk > 1 ? workspace : dw;
  hipStream_t st = (hipStream_t)stream;
  int bc = pick_tile_dim(co_write), bd = pick_tile_dim(ci_write);
  if (tile_override() == 64) { bc = 64; bd = 64; }
  int rc;
  if (bc == 128 && bd == 128) rc = launch_wgrad<128, 128>(a, taps, splitk, prec, st);
  else if (bc == 128) rc = launch_wgrad<128, 64>(a, taps, splitk, prec, st);
  else if (bd == 128) rc = launch_wgrad<64, 128>(a, taps, splitk, prec, st);
  else rc = launch_wgrad<64, 64>(a, taps, splitk, prec, st);
  if (rc) return rc;
  if (splitk > 1) {
    long n = a.slab;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, n, splitk, a.slab);
    rc = ZS3_LAUNCH_CHECK();
  }
  return rc;
}
