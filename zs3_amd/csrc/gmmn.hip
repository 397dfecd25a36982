// The GMMN generator's per-(image, class) update as latency-shaped kernels (train_pascal_GMMN.py:209-242).
//
// One update runs the two-layer MLP (gmmn.py:17-22: Linear 600->256, LeakyReLU(0.2), Dropout(0.5), Linear 256->256) on
// S = 128 sampled rows, the MMD loss, the MLP backward and one Adam step, and the NEXT update needs the new weights: a
// dependent chain of ~10 stages of a few MFLOP each, 225 times per training step.  Nothing here is throughput-bound; what
// counts is the length of the chain and the latency of every stage.  The general conv kernels spent 10-13 us per stage
// (19 dependent K steps on 8 workgroups); these kernels are shaped for the problem instead:
//   * mlp_gemm_kernel: a row-GEMM out[M][N] = A[M][K] * W[N][K]^T on 32 x 16 output tiles (64 workgroups for 128 x 256):
//     the workgroup loads its WHOLE K extent of both operands in one burst (one memory round trip), its four waves split
//     K between them (bf16x3 on v_mfma_f32_16x16x32_bf16, fp32 accumulate) and combine through LDS in a fixed order.
//     Fused around it: the gather of the class-embedding rows + the noise draw (forward 1), bias + LeakyReLU + Dropout
//     (forward 1), bias + the gather of the sampled real features (forward 2), Dropout-backward * LeakyReLU' (dgrad).
//   * mlp_wgrad_kernel: both layers' weight gradients dW[co][ci] = sum_r dY[r][co] X[r][ci] (reduction over the 128 rows)
//     and both bias gradients in ONE launch of 64 x 64 tiles.
// 16 launches per update became 8 (round 2), 7 with Adam inside the weight-gradient launch (round 3), 6 with the table-driven
// first GEMM (round 4: zs3_gmmn_mlp_fwd1_table).  Arithmetic: every product is the bf16x3 split of common.h.
#include "common.h"
#include "zs3hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f4;

// eight fp32 values -> the bf16 hi / lo fragments of one MFMA operand (bf16x3 split, common.h)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split_pair<3>(v[2 * e], v[2 * e + 1], h[e], l[e]);
  const u32x4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
  hi = __builtin_bit_cast(bf16x8, hv);
  lo = __builtin_bit_cast(bf16x8, lv);
}

enum { MLP_FWD1 = 1, MLP_FWD2 = 2, MLP_DGRAD = 3 };

struct MlpArgs {
  const float* a;            // [M][lda] fp32 (FWD2: hd, DGRAD: d(gen)); unused by FWD1
  const unsigned short* w;   // packed planes of zs3_prep_weight: row n at w + n * kchunks * 64
  const float* bias;         // [N] or null
  float* out;                // [M][ldo]: FWD1 h (after LeakyReLU), FWD2 gen, DGRAD d(pre-activation)
  float* out2;               // FWD1: hd = dropout(h), [M][ldo]
  int lda, ldo, M, N, K, kchunks;
  // FWD1: A row r = [emb[pix[r]][0:Ca] | U[0,1)^Cb keyed on key[r] | 0]; the rows are also stored to x_out (wgrad operand)
  const float* emb;
  const long* pix;
  const long* key;           // sampled within-class pixel index: keys the noise (FWD1) and the dropout mask (FWD1, DGRAD)
  float* x_out;
  int ld_emb, Ca, Cb, ldx;
  // FWD2: real_out[r] = real[gidx[r]] (the MMD's real samples), copied by the workgroups of this launch
  const float* real;
  const long* gidx;
  float* real_out;
  int ld_real;
  // DGRAD: h of FWD1 (LeakyReLU mask)
  const float* h;
  int ldh;
  float leak, p_drop;
  unsigned long long seed_noise, seed_drop;
  const unsigned long long* seed_dev;
  // FWD1, table-driven (zs3_gmmn_mlp_fwd1_table: the work of gmmn_prep_kernel inside this launch): row u = upd[0] of `table`
  // ([S sample indices | order offset | pixel base]) selects the update's (image, class); pix / key above are ignored
  const long* table;
  const long* upd;
  const long* order;
  long* pix_out;             // [S] row of the real features of every sample (FWD2's gather list)
  long* key_out;             // [S] the sample indices as a plain array (DGRAD's dropout keys)
  const long* adam_step;
  float* adam_bc;
  float b1, b2;
  int ld_table, S;
};

constexpr int MLP_BM = 32, MLP_BN = 16, MLP_MAXCH = 20;   // <= 640 reduction elements per call

template <int MODE>
__global__ __launch_bounds__(256) void mlp_gemm_kernel(const MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int kpad = p.kchunks * 32;
  const int SA = kpad + 4;                    // floats per A row (+16 B: ds_read_b128 of 16 rows hits 16 bank quads)
  const int SB = p.kchunks * 64 + 8;          // bf16 per W row (+16 B)
  float* As = reinterpret_cast<float*>(smem);
  unsigned short* Bs = reinterpret_cast<unsigned short*>(smem + (size_t)MLP_BM * SA * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = (p.N + MLP_BN - 1) / MLP_BN;
  const int mb = blockIdx.x / nb, nbk = blockIdx.x - mb * nb;
  const int m0 = mb * MLP_BM, n0 = nbk * MLP_BN;
  unsigned long long s_noise = p.seed_noise, s_drop = p.seed_drop;
  if (p.seed_dev) {
    s_noise += p.seed_dev[0];
    s_drop += p.seed_dev[0];
  }

  // ---- one burst: the whole K extent of this tile's rows of A and of W.  Every global load is issued before the first
  // LDS write (and before the noise is hashed), so the tile costs ONE memory round trip; <= 20 K chunks (checked by the host)
  // thread t owns row t >> 3 of A (8 threads x 16 B walk a row in 128-byte steps) and row t >> 4 of W (16 threads per
  // row): no integer division in the address math, every wave instruction reads whole 128-byte lines
  constexpr int MAXA = (MLP_MAXCH * 8 + 7) / 8, MAXB = (MLP_MAXCH * 8 + 15) / 16;
  const int a4 = kpad >> 2;                    // float4 per A row
  const int b16 = p.kchunks * 8;               // 16-byte pieces per W row
  const int ar = tid >> 3, aq = tid & 7, br = tid >> 4, bq = tid & 15;
  const int am = m0 + ar;
  f4 av[MAXA];
  u32x4 bv[MAXB];
  const unsigned short* wrow = p.w + (size_t)(n0 + br) * p.kchunks * 64;
  const bool wok = n0 + br < p.N;
#pragma unroll
  for (int u = 0; u < MAXB; ++u) {
    const int piece = bq + 16 * u;
    bv[u] = u32x4{0u, 0u, 0u, 0u};
    if (wok && piece < b16) bv[u] = *reinterpret_cast<const u32x4*>(wrow + piece * 8);
  }
  const float* arow = nullptr;
  const long* keys = p.key;
  if (MODE == MLP_FWD1 && p.table) {   // table-driven: the update's row of the step's device table holds the sample indices
    const long* row = p.table + p.upd[0] * p.ld_table;
    keys = row;
    if (am < p.M) {
      const long pixg = row[p.S + 1] + p.order[row[p.S] + row[am]];
      arow = p.emb + pixg * p.ld_emb;
      if (nbk == 0 && aq == 0) {
        p.pix_out[am] = pixg;
        p.key_out[am] = row[am];
      }
    }
    if (p.adam_bc && blockIdx.x == 0 && tid == 0) {   // Adam's bias corrections of THIS update (torch: 1 - beta ** step in doubles)
      const double st = (double)(p.adam_step[0] + 1);
      p.adam_bc[0] = (float)(1.0 - pow((double)p.b1, st));
      p.adam_bc[1] = (float)sqrt(1.0 - pow((double)p.b2, st));
    }
  } else if (am < p.M)   // FWD1 without a row list reads row am itself (x already assembled by zs3_gmmn_prep: no dependent index load)
    arow = MODE == MLP_FWD1 ? p.emb + (p.pix ? p.pix[am] : (long)am) * p.ld_emb : p.a + (size_t)am * p.lda;
  const int alim = MODE == MLP_FWD1 ? p.Ca : p.K;
#pragma unroll
  for (int u = 0; u < MAXA; ++u) {
    const int c = (aq + 8 * u) * 4;
    av[u] = f4{0.f, 0.f, 0.f, 0.f};
    if (arow && c < alim) av[u] = *reinterpret_cast<const f4*>(arow + c);
  }
  if (MODE == MLP_FWD1 && am < p.M) {
#pragma unroll
    for (int u = 0; u < MAXA; ++u) {
      const int c = (aq + 8 * u) * 4;
      if (c >= p.Ca && c < p.Ca + p.Cb) {
        const unsigned long long base = (unsigned long long)(keys[am] * p.Cb + (c - p.Ca));
#pragma unroll
        for (int e = 0; e < 4; ++e) av[u][e] = u01(s_noise, base + e);
      }
      if (p.x_out && nbk == 0 && c < p.ldx && c < 4 * a4) *reinterpret_cast<f4*>(p.x_out + (size_t)am * p.ldx + c) = av[u];
    }
  }
#pragma unroll
  for (int u = 0; u < MAXA; ++u) {
    const int c4i = aq + 8 * u;
    if (c4i < a4) *reinterpret_cast<f4*>(As + ar * SA + c4i * 4) = av[u];
  }
#pragma unroll
  for (int u = 0; u < MAXB; ++u) {
    const int piece = bq + 16 * u;
    if (piece < b16) *reinterpret_cast<u32x4*>(Bs + br * SB + piece * 8) = bv[u];
  }
  if (MODE == MLP_FWD2 && p.real_out) {       // the sampled real rows, 16 columns per workgroup
    for (int i = tid; i < MLP_BM * (MLP_BN / 4); i += 256) {
      const int r = i / (MLP_BN / 4), c = n0 + (i - r * (MLP_BN / 4)) * 4;
      const int m = m0 + r;
      if (m < p.M && c < p.N)
        *reinterpret_cast<f4*>(p.real_out + (size_t)m * p.N + c) = *reinterpret_cast<const f4*>(p.real + p.gidx[m] * p.ld_real + c);
    }
  }
  __syncthreads();

  // ---- wave w multiplies the K chunks w, w+4, ...: two 16x16 tiles (rows 0-15 / 16-31), bf16x3
  const int r16 = lane & 15, kg = lane >> 4;
  f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int ch = wave; ch < p.kchunks; ch += 4) {
    const bf16x8 b_hi = *reinterpret_cast<const bf16x8*>(Bs + r16 * SB + ch * 64 + kg * 8);
    const bf16x8 b_lo = *reinterpret_cast<const bf16x8*>(Bs + r16 * SB + ch * 64 + 32 + kg * 8);
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
      const float* src = As + (rf * 16 + r16) * SA + ch * 32 + kg * 8;
      const f4 v0 = *reinterpret_cast<const f4*>(src), v1 = *reinterpret_cast<const f4*>(src + 4);
      const float av[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      bf16x8 a_hi, a_lo;
      split8(av, a_hi, a_lo);
      acc[rf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, acc[rf], 0, 0, 0);
      acc[rf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, acc[rf], 0, 0, 0);
      acc[rf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, acc[rf], 0, 0, 0);
    }
  }
  __syncthreads();   // operands are no longer read: the partial tiles reuse the A region
  f4* red = reinterpret_cast<f4*>(smem);      // [wave][rf][lane]
  red[(wave * 2 + 0) * 64 + lane] = acc[0];
  red[(wave * 2 + 1) * 64 + lane] = acc[1];
  __syncthreads();
  if (tid >= 128) return;
  const int rf = tid >> 6;
  f4 v = red[(0 * 2 + rf) * 64 + lane];
#pragma unroll
  for (int w = 1; w < 4; ++w) v += red[(w * 2 + rf) * 64 + lane];   // fixed order: deterministic
  const int col = n0 + r16;
  if (col >= p.N) return;
  // everything the four rows of this lane need from memory in one burst of independent (clamped, unbranched) loads: a
  // load -> use -> store chain per row costs an L2 round trip per row (the DGRAD epilogue was eight of them)
  const float bias_v = p.bias ? p.bias[col] : 0.f;
  const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
  const bool masks = MODE != MLP_FWD2 && p.p_drop > 0.f;
  long key_v[4];
  float h_v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = m0 + rf * 16 + kg * 4 + e;   // C layout of the 16x16 MFMA: row = (lane >> 4) * 4 + e, col = lane & 15
    const int mc = m < p.M ? m : 0;
    key_v[e] = masks ? keys[mc] : 0;
    h_v[e] = MODE == MLP_DGRAD ? p.h[(size_t)mc * p.ldh + col] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = m0 + rf * 16 + kg * 4 + e;
    if (m >= p.M) continue;
    float t = v[e] + bias_v;
    if (MODE == MLP_FWD1) {
      t = t > 0.f ? t : t * p.leak;
      p.out[(size_t)m * p.ldo + col] = t;
      if (masks) t = u01(s_drop, (unsigned long long)(key_v[e] * p.N + col)) >= p.p_drop ? t * inv_keep : 0.f;
      p.out2[(size_t)m * p.ldo + col] = t;
    } else if (MODE == MLP_FWD2) {
      p.out[(size_t)m * p.ldo + col] = t;
    } else {
      if (masks) t = u01(s_drop, (unsigned long long)(key_v[e] * p.N + col)) >= p.p_drop ? t * inv_keep : 0.f;
      p.out[(size_t)m * p.ldo + col] = h_v[e] > 0.f ? t : t * p.leak;
    }
  }
}

template <int MODE>
int launch_mlp(const MlpArgs& a, hipStream_t st) {
  if (a.kchunks < 1 || a.kchunks > MLP_MAXCH) return -2;
  const size_t lds = (size_t)MLP_BM * (a.kchunks * 32 + 4) * 4 + (size_t)MLP_BN * (a.kchunks * 64 + 8) * 2;
  static size_t configured = 0;
  if (lds > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_gemm_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return -4;
    configured = lds;
  }
  const int blocks = ((a.M + MLP_BM - 1) / MLP_BM) * ((a.N + MLP_BN - 1) / MLP_BN);
  hipLaunchKernelGGL((mlp_gemm_kernel<MODE>), dim3(blocks), dim3(256), lds, st, a);
  return ZS3_LAUNCH_CHECK();
}

// ---- both weight gradients and both bias gradients of the MLP in one launch -----------------------------------------
struct WgProb {
  const float* dy;   // [R][ldy], co columns
  const float* x;    // [R][ldx], ci columns
  float* dw;         // [co][ci] (row stride ci)
  float* db;         // [co]
  int ldy, ldx, co, ci, tiles_ci, tile0;   // tile0: first block id of this problem
  // fused Adam (mlp_wgrad_kernel<true>): the layer's weight [co][ci] and bias [co] with their moments, and the bf16 hi/lo
  // operand planes of zs3_prep_weight that the next forward / dgrad read (f_pk rows = co, t_pk rows = ci)
  float *w, *wm, *wv, *b, *bm, *bv;
  unsigned short *f_pk, *t_pk;
  int cin_pad, cout_pad;
};
struct WgArgs {
  WgProb pr[2];
  int R;
  // fused Adam + end-of-update bookkeeping
  float lr, b1, b2, eps, wd;
  const long* step;        // Adam step count before this update
  const float* bc;         // {1 - b1^t, sqrt(1 - b2^t)} of this update from zs3_gmmn_prep, or null (computed here)
  long* counters[3];       // slot (update index of the step), step, seed: advanced by the last workgroup to finish
  long seed_inc;
  unsigned* done;          // arrival counter (zero between launches)
};

__device__ __forceinline__ long packed_index(long row, long k, long ktot, int half) {   // layout of zs3_prep_weight, 1 tap
  return ((row * (ktot >> 5) + (k >> 5)) * 2 + half) * 32 + (k & 31);
}
__device__ __forceinline__ float adam_update(float g, float& p, float& m, float& v, float lr, float b1, float b2, float eps,
                                             float wd, float bc1, float bc2_sqrt) {   // torch.optim.Adam, as zs3_adam_multi
  const float gi = g + wd * p;
  m = m + (1.f - b1) * (gi - m);
  v = b2 * v + (1.f - b2) * gi * gi;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - (lr / bc1) * (m / denom);
  return p;
}

constexpr int WG_T = 64, WG_LD = 66;   // 66-float rows: the two 8-row halves of a ds_read_b32 group land 16 banks apart

template <bool ADAM>
__global__ __launch_bounds__(256) void mlp_wgrad_kernel(const WgArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x 33 KB: above the 64 KB static limit
  float* Ys = reinterpret_cast<float*>(smem);
  float* Xs = Ys + 128 * WG_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const WgProb& q = p.pr[blockIdx.x >= p.pr[1].tile0 ? 1 : 0];
  const int t = blockIdx.x - q.tile0;
  const int tco = t / q.tiles_ci, tci = t - tco * q.tiles_ci;
  const int co0 = tco * WG_T, ci0 = tci * WG_T;
  const int wi = wave >> 1, wj = wave & 1;
  const int m16 = lane & 15, kg = lane >> 4;
  // ONE burst of independent loads: the 16 staging loads of the thread and (fused Adam) its 4 x 4 weights and moments are all
  // in flight before the first LDS store.  Branch-free on purpose: out-of-range rows / columns read row 0 / column 0 and are
  // zeroed by a select afterwards -- with the loads under `if`s the compiler waited for each one before issuing the next
  // (16 dependent L2 round trips: 10 of this kernel's 18 us).
  const int c = (tid & 15) * 4, r0 = tid >> 4;      // 16 threads per row, 16 rows per pass
  const bool cy_ok = co0 + c < q.co, cx_ok = ci0 + c < q.ci;   // co, ci: multiples of 4
  const float* ybase = q.dy + (cy_ok ? co0 + c : 0);
  const float* xbase = q.x + (cx_ok ? ci0 + c : 0);
  f4 vy[8], vx[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = r0 + it * 16;
    const size_t rr = r < p.R ? (size_t)r : 0;
    vy[it] = *reinterpret_cast<const f4*>(ybase + rr * q.ldy);
    vx[it] = *reinterpret_cast<const f4*>(xbase + rr * q.ldx);
  }
  f4 w_old[2][2], m_old[2][2], v_old[2][2];
  if (ADAM) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = co0 + wi * 32 + i * 16 + m16, ci = ci0 + wj * 32 + j * 16 + kg * 4;
        const size_t idx = (co < q.co && ci < q.ci) ? (size_t)co * q.ci + ci : 0;   // out of range: element 0, never used
        w_old[i][j] = *reinterpret_cast<const f4*>(q.w + idx);
        m_old[i][j] = *reinterpret_cast<const f4*>(q.wm + idx);
        v_old[i][j] = *reinterpret_cast<const f4*>(q.wv + idx);
      }
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = r0 + it * 16;
    const bool r_ok = r < p.R;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Ys[r * WG_LD + c + e] = (r_ok && cy_ok) ? vy[it][e] : 0.f;
      Xs[r * WG_LD + c + e] = (r_ok && cx_ok) ? vx[it][e] : 0.f;
    }
  }
  __syncthreads();
  f4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < 4; ++kc) {          // 32 rows of the reduction per MFMA
    const int k0 = kc * 32 + kg * 8;
    bf16x8 a_hi[2], a_lo[2], b_hi[2], b_lo[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float ya[8], xb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ya[e] = Ys[(k0 + e) * WG_LD + wi * 32 + f * 16 + m16];
        xb[e] = Xs[(k0 + e) * WG_LD + wj * 32 + f * 16 + m16];
      }
      split8(ya, a_hi[f], a_lo[f]);
      split8(xb, b_hi[f], b_lo[f]);
    }
    // the x fragment is the MFMA's row operand: a lane then holds FOUR CONSECUTIVE ci of one co (C layout: row =
    // (lane >> 4) * 4 + e, col = lane & 15), i.e. 16 contiguous bytes of dw / weight / moment rows and 8 of the f_pk plane
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_lo[j], a_hi[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hi[j], a_lo[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hi[j], a_hi[i], acc[i][j], 0, 0, 0);
      }
  }
  float bc1 = 1.f, bc2_sqrt = 1.f;
  if (ADAM) {
    if (p.bc) {           // computed once per update by zs3_gmmn_prep
      bc1 = p.bc[0];
      bc2_sqrt = p.bc[1];
    } else {
      const double st = (double)(p.step[0] + 1);
      bc1 = (float)(1.0 - pow((double)p.b1, st));
      bc2_sqrt = (float)sqrt(1.0 - pow((double)p.b2, st));
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = co0 + wi * 32 + i * 16 + m16;
      const int ci = ci0 + wj * 32 + j * 16 + kg * 4;   // ci .. ci + 3 (channel counts are multiples of 4)
      if (co >= q.co || ci >= q.ci) continue;
      const size_t idx = (size_t)co * q.ci + ci;
      if (!ADAM) {
        *reinterpret_cast<f4*>(q.dw + idx) = acc[i][j];
        continue;
      }
      // the tile's gradient never leaves the registers: Adam on four weights, then their bf16 hi/lo planes
      f4 w = w_old[i][j], m = m_old[i][j], v = v_old[i][j];
      unsigned short h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float we = w[e], me = m[e], ve = v[e];
        adam_update(acc[i][j][e], we, me, ve, p.lr, p.b1, p.b2, p.eps, p.wd, bc1, bc2_sqrt);
        w[e] = we; m[e] = me; v[e] = ve;
        h[e] = f32_to_bf16_rne(we);
        l[e] = f32_to_bf16_rne(we - bf16_bits_to_f32(h[e]));
      }
      *reinterpret_cast<f4*>(q.w + idx) = w;
      *reinterpret_cast<f4*>(q.wm + idx) = m;
      *reinterpret_cast<f4*>(q.wv + idx) = v;
      const u32x2 hp = {(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
      const u32x2 lp = {(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
      *reinterpret_cast<u32x2*>(q.f_pk + packed_index(co, ci, q.cin_pad, 0)) = hp;   // 4 consecutive k of one 32-chunk
      *reinterpret_cast<u32x2*>(q.f_pk + packed_index(co, ci, q.cin_pad, 1)) = lp;
      if (q.t_pk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          q.t_pk[packed_index(ci + e, co, q.cout_pad, 0)] = h[e];
          q.t_pk[packed_index(ci + e, co, q.cout_pad, 1)] = l[e];
        }
      }
    }
  if (tci == 0 && tid < WG_T && co0 + tid < q.co) {   // bias gradient: column sums of dy, rows in order
    float s = 0.f;   // rows in order (one chain: the order is part of the result)
#pragma unroll 16
    for (int r = 0; r < p.R; ++r) s += Ys[r * WG_LD + tid];
    if (!ADAM) {
      if (q.db) q.db[co0 + tid] = s;
    } else if (q.b) {
      float w = q.b[co0 + tid], m = q.bm[co0 + tid], v = q.bv[co0 + tid];
      adam_update(s, w, m, v, p.lr, p.b1, p.b2, p.eps, p.wd, bc1, bc2_sqrt);
      q.b[co0 + tid] = w;
      q.bm[co0 + tid] = m;
      q.bv[co0 + tid] = v;
    }
  }
  if (ADAM) {
    // end of the update: the LAST workgroup to get here advances the device-resident counters (every workgroup has read
    // the step count by then; the readers of slot and seed are other launches, ordered by the stream)
    __syncthreads();
    if (tid == 0) {
      const unsigned n = __hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (n == gridDim.x - 1) {
        __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p.counters[0][0] += 1;
        p.counters[1][0] += 1;
        p.counters[2][0] += p.seed_inc;
      }
    }
  }
}

// ---- first launch of a table-driven update: everything the update needs from its (image, class) is read from a device table
// row selected by a device-resident counter, so a captured update replays without any host-side argument.
// table row u = [ridx[0..S) | order offset | pixel base]: sample j is the ridx[j]-th pixel of the class segment that starts at
// order[order offset]; x[j] = [emb[base + pixel] | U[0,1)^Cb keyed on ridx[j]]; pix_global[j] = base + pixel (row of the real
// features), key[j] = ridx[j] (dropout key).  One workgroup per sampled row.
__global__ __launch_bounds__(256) void gmmn_prep_kernel(const long* table, int ld_table, const long* upd, const long* order,
                                                        const float* emb, int ld_emb, int Ca, int Cb, float* x, int ldx,
                                                        long* pix_global, long* key, int S, unsigned long long seed,
                                                        const unsigned long long* seed_dev, const long* adam_step, float b1,
                                                        float b2, float* adam_bc) {
  // Adam's bias corrections of THIS update (torch: 1 - beta ** step in Python doubles): one thread of this latency-bound
  // launch computes them for the 14 k threads of the weight-gradient launch, where the two double-precision pow() calls per
  // thread were 10 % of the instruction stream
  if (adam_bc && blockIdx.x == 0 && threadIdx.x == 0) {
    const double st = (double)(adam_step[0] + 1);
    adam_bc[0] = (float)(1.0 - pow((double)b1, st));
    adam_bc[1] = (float)sqrt(1.0 - pow((double)b2, st));
  }
  const long* row = table + upd[0] * ld_table;
  const int j = blockIdx.x;
  const long r = row[j], base = row[S + 1];
  const long pixel = order[row[S] + r];
  if (threadIdx.x == 0) {
    pix_global[j] = base + pixel;
    key[j] = r;
  }
  if (seed_dev) seed += seed_dev[0];
  const float* src = emb + (base + pixel) * ld_emb;
  for (int c = threadIdx.x * 4; c < Ca + Cb; c += 256 * 4) {
    f4 v;
    if (c < Ca) {
      v = *reinterpret_cast<const f4*>(src + c);
    } else {
      const unsigned long long b0 = (unsigned long long)(r * Cb + (c - Ca));
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = u01(seed, b0 + e);
    }
    *reinterpret_cast<f4*>(x + (size_t)j * ldx + c) = v;
  }
}

}  // namespace

extern "C" int zs3_gmmn_prep(const long* table, int ld_table, const void* upd_dev, const long* order, const float* emb,
                             int ld_emb, int Ca, int Cb, float* x, int ldx, long* pix_global, long* key, int S,
                             unsigned long long seed, const void* seed_dev, const void* adam_step_dev, float b1, float b2,
                             float* adam_bc, void* stream) {
  if (S <= 0) return 0;
  if ((Ca & 3) || (Cb & 3) || (ld_emb & 3) || (ldx & 3) || ld_table < S + 2 || !upd_dev || (adam_bc && !adam_step_dev)) return -1;
  hipLaunchKernelGGL(gmmn_prep_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, table, ld_table, (const long*)upd_dev, order,
                     emb, ld_emb, Ca, Cb, x, ldx, pix_global, key, S, seed, (const unsigned long long*)seed_dev,
                     (const long*)adam_step_dev, b1, b2, adam_bc);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_gmmn_mlp_fwd1(const float* emb, int ld_emb, const long* pix, const long* key, int Ca, int Cb,
                                 const void* w_pk, int kchunks, const float* bias, float* x_out, int ldx, float* h,
                                 float* hd, int ldo, int M, int N, float leak, float p_drop, unsigned long long seed_noise,
                                 unsigned long long seed_drop, const void* seed_dev, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if ((Ca & 3) || (Cb & 3) || (ld_emb & 3) || (ldx & 3) || Ca + Cb > kchunks * 32 || (x_out && ldx > kchunks * 32) || !key) return -1;   // pix == NULL: rows in place
  MlpArgs a = {};
  a.w = (const unsigned short*)w_pk; a.bias = bias; a.out = h; a.out2 = hd; a.ldo = ldo; a.M = M; a.N = N; a.K = Ca + Cb;
  a.kchunks = kchunks; a.emb = emb; a.pix = pix; a.key = key; a.x_out = x_out; a.ld_emb = ld_emb; a.Ca = Ca; a.Cb = Cb;
  a.ldx = ldx; a.leak = leak; a.p_drop = p_drop; a.seed_noise = seed_noise; a.seed_drop = seed_drop;
  a.seed_dev = (const unsigned long long*)seed_dev;
  return launch_mlp<MLP_FWD1>(a, (hipStream_t)stream);
}

extern "C" int zs3_gmmn_mlp_fwd1_table(const long* table, int ld_table, const void* upd_dev, const long* order, const float* emb,
                                       int ld_emb, int Ca, int Cb, const void* w_pk, int kchunks, const float* bias, float* x_out,
                                       int ldx, float* h, float* hd, int ldo, int S, int N, float leak, float p_drop,
                                       unsigned long long seed_noise, unsigned long long seed_drop, const void* seed_dev,
                                       long* pix_global, long* key, const void* adam_step_dev, float b1, float b2, float* adam_bc,
                                       void* stream) {
  if (S <= 0 || N <= 0) return 0;
  if ((Ca & 3) || (Cb & 3) || (ld_emb & 3) || (ldx & 3) || Ca + Cb > kchunks * 32 || (x_out && ldx > kchunks * 32) || ld_table < S + 2 ||
      !table || !upd_dev || !order || !pix_global || !key || (adam_bc && !adam_step_dev))
    return -1;
  MlpArgs a = {};
  a.w = (const unsigned short*)w_pk; a.bias = bias; a.out = h; a.out2 = hd; a.ldo = ldo; a.M = S; a.N = N; a.K = Ca + Cb;
  a.kchunks = kchunks; a.emb = emb; a.x_out = x_out; a.ld_emb = ld_emb; a.Ca = Ca; a.Cb = Cb;
  a.ldx = ldx; a.leak = leak; a.p_drop = p_drop; a.seed_noise = seed_noise; a.seed_drop = seed_drop;
  a.seed_dev = (const unsigned long long*)seed_dev;
  a.table = table; a.ld_table = ld_table; a.upd = (const long*)upd_dev; a.order = order; a.pix_out = pix_global; a.key_out = key;
  a.S = S; a.adam_step = (const long*)adam_step_dev; a.b1 = b1; a.b2 = b2; a.adam_bc = adam_bc;
  return launch_mlp<MLP_FWD1>(a, (hipStream_t)stream);
}

extern "C" int zs3_gmmn_mlp_fwd2(const float* hd, int lda, const void* w_pk, int kchunks, const float* bias, float* gen,
                                 int ldo, int M, int N, int K, const float* real, int ld_real, const long* gidx,
                                 float* real_out, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if ((lda & 3) || (K & 3) || K > kchunks * 32 || (real_out && ((N & 3) || (ld_real & 3) || !real || !gidx))) return -1;
  MlpArgs a = {};
  a.a = hd; a.lda = lda; a.w = (const unsigned short*)w_pk; a.kchunks = kchunks; a.bias = bias; a.out = gen; a.ldo = ldo;
  a.M = M; a.N = N; a.K = K; a.real = real; a.ld_real = ld_real; a.gidx = gidx; a.real_out = real_out;
  return launch_mlp<MLP_FWD2>(a, (hipStream_t)stream);
}

extern "C" int zs3_gmmn_mlp_dgrad(const float* dgen, int lda, const void* wt_pk, int kchunks, const float* h, int ldh,
                                  const long* key, float* dpre, int ldo, int M, int N, int K, float leak, float p_drop,
                                  unsigned long long seed_drop, const void* seed_dev, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if ((lda & 3) || (K & 3) || K > kchunks * 32 || !h || (p_drop > 0.f && !key)) return -1;
  MlpArgs a = {};
  a.a = dgen; a.lda = lda; a.w = (const unsigned short*)wt_pk; a.kchunks = kchunks; a.out = dpre; a.ldo = ldo; a.M = M;
  a.N = N; a.K = K; a.h = h; a.ldh = ldh; a.key = key; a.leak = leak; a.p_drop = p_drop; a.seed_drop = seed_drop;
  a.seed_dev = (const unsigned long long*)seed_dev;
  return launch_mlp<MLP_DGRAD>(a, (hipStream_t)stream);
}

extern "C" int zs3_gmmn_mlp_wgrad(const float* dy2, int ldy2, const float* x2, int ldx2, int co2, int ci2, float* dw2,
                                  float* db2, const float* dy1, int ldy1, const float* x1, int ldx1, int co1, int ci1,
                                  float* dw1, float* db1, int R, void* stream) {
  if (R <= 0 || R > 128) return -1;
  if ((ldy2 | ldx2 | co2 | ci2 | ldy1 | ldx1 | co1 | ci1) & 3) return -1;
  WgArgs a = {};
  a.R = R;
  const float* dys[2] = {dy2, dy1};
  const float* xs[2] = {x2, x1};
  float* dws[2] = {dw2, dw1};
  float* dbs[2] = {db2, db1};
  const int ldys[2] = {ldy2, ldy1}, ldxs[2] = {ldx2, ldx1}, cos_[2] = {co2, co1}, cis[2] = {ci2, ci1};
  int tiles = 0;
  for (int i = 0; i < 2; ++i) {
    WgProb& q = a.pr[i];
    q.dy = dys[i]; q.x = xs[i]; q.dw = dws[i]; q.db = dbs[i]; q.ldy = ldys[i]; q.ldx = ldxs[i]; q.co = cos_[i]; q.ci = cis[i];
    q.tiles_ci = (cis[i] + WG_T - 1) / WG_T;
    q.tile0 = tiles;
    tiles += q.tiles_ci * ((cos_[i] + WG_T - 1) / WG_T);
  }
  constexpr int LDS = 2 * 128 * WG_LD * 4;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_wgrad_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
        hipSuccess)
      return -4;
    configured = true;
  }
  hipLaunchKernelGGL(mlp_wgrad_kernel<false>, dim3(tiles), dim3(256), LDS, (hipStream_t)stream, a);
  return ZS3_LAUNCH_CHECK();
}

/* zs3_gmmn_mlp_wgrad + Adam in one launch: state[l] = {w, exp_avg, exp_avg_sq, bias, bias exp_avg, bias exp_avg_sq, f_pk,
 * t_pk} (8 device pointers per layer, layer 2 first), dims[l] = {cin_pad, cout_pad}.  The last workgroup to finish
 * advances slot_dev, step_dev (+1 each) and seed_dev (+seed_inc); done_dev is a zeroed 32-bit arrival counter. */
extern "C" int zs3_gmmn_mlp_wgrad_adam(const float* dy2, int ldy2, const float* x2, int ldx2, int co2, int ci2,
                                       const float* dy1, int ldy1, const float* x1, int ldx1, int co1, int ci1, int R,
                                       const void* const* state2, const void* const* state1, int cin_pad2, int cout_pad2,
                                       int cin_pad1, int cout_pad1, float lr, float b1, float b2, float eps, float wd,
                                       void* slot_dev, void* step_dev, void* seed_dev, long seed_inc, void* done_dev,
                                       const float* adam_bc, void* stream) {
  if (R <= 0 || R > 128 || !state2 || !state1 || !slot_dev || !step_dev || !seed_dev || !done_dev) return -1;
  if ((ldy2 | ldx2 | co2 | ci2 | ldy1 | ldx1 | co1 | ci1) & 3) return -1;
  WgArgs a = {};
  a.R = R;
  const float* dys[2] = {dy2, dy1};
  const float* xs[2] = {x2, x1};
  const void* const* sts[2] = {state2, state1};
  const int ldys[2] = {ldy2, ldy1}, ldxs[2] = {ldx2, ldx1}, cos_[2] = {co2, co1}, cis[2] = {ci2, ci1};
  const int cips[2] = {cin_pad2, cin_pad1}, cops[2] = {cout_pad2, cout_pad1};
  int tiles = 0;
  for (int i = 0; i < 2; ++i) {
    WgProb& q = a.pr[i];
    q.dy = dys[i]; q.x = xs[i]; q.ldy = ldys[i]; q.ldx = ldxs[i]; q.co = cos_[i]; q.ci = cis[i];
    q.w = (float*)sts[i][0]; q.wm = (float*)sts[i][1]; q.wv = (float*)sts[i][2];
    q.b = (float*)sts[i][3]; q.bm = (float*)sts[i][4]; q.bv = (float*)sts[i][5];
    q.f_pk = (unsigned short*)sts[i][6]; q.t_pk = (unsigned short*)sts[i][7];
    if (!q.w || !q.wm || !q.wv || !q.f_pk) return -1;
    q.cin_pad = cips[i]; q.cout_pad = cops[i];
    q.tiles_ci = (cis[i] + WG_T - 1) / WG_T;
    q.tile0 = tiles;
    tiles += q.tiles_ci * ((cos_[i] + WG_T - 1) / WG_T);
  }
  a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = wd;
  a.step = (const long*)step_dev;
  a.bc = adam_bc;
  a.counters[0] = (long*)slot_dev; a.counters[1] = (long*)step_dev; a.counters[2] = (long*)seed_dev;
  a.seed_inc = seed_inc;
  a.done = (unsigned*)done_dev;
  constexpr int LDS = 2 * 128 * WG_LD * 4;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_wgrad_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
        hipSuccess)
      return -4;
    configured = true;
  }
  hipLaunchKernelGGL(mlp_wgrad_kernel<true>, dim3(tiles), dim3(256), LDS, (hipStream_t)stream, a);
  return ZS3_LAUNCH_CHECK();
}
