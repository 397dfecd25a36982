// Losses of the ZS3 hot path on gfx950:
//   * weighted cross-entropy with ignore_index over NHWC logits (zs3/utils/loss.py:31-46) fwd/bwd,
//   * multi-bandwidth Gaussian MMD of the GMMN (zs3/utils/loss.py:92-115) fwd/bwd, Gram matrices on the
//     exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32) because E_ij = <xi,xj> - |xi|^2/2 - |xj|^2/2 cancels.
#include "common.h"
#include "zs3hip.h"

namespace {

// ------------------------------------------------------------------------------------------------ CE
template <typename TT>
__device__ __forceinline__ int load_target(const TT* t, long i) { return (int)(long)t[i]; }

// One block = 256 consecutive pixels.  Their logits ([256][C] floats, contiguous when ld == C) are staged through
// LDS with fully coalesced loads / stores; thread t then owns pixel t (row stride C+1 => conflict-free).
template <typename TT, bool BWD>
__global__ __launch_bounds__(256) void ce_tile_kernel(const float* logits, int ld, const TT* target, const float* weight,
                                                     long P, int C, int ignore_index, double* partial,
                                                     const float* loss_ws, const float* gout, float inv_batch,
                                                     float* dlogits, int ldo) {
  extern __shared__ float tile[];  // [256][C+1]
  const int CP = C + 1;
  const float coef = BWD ? gout[0] * inv_batch / loss_ws[1] : 0.f;
  double lsum = 0.0, wsum = 0.0;
  for (long p0 = (long)blockIdx.x * 256; p0 < P; p0 += (long)gridDim.x * 256) {
    const int np = (int)min((long)256, P - p0);
    const bool dense = (ld == C);
    if (dense) {
      const float* src = logits + p0 * C;
      for (int e = threadIdx.x; e < np * C; e += 256) tile[(e / C) * CP + (e % C)] = src[e];
    } else {
      for (int e = threadIdx.x; e < np * C; e += 256) tile[(e / C) * CP + (e % C)] = logits[(p0 + e / C) * ld + (e % C)];
    }
    __syncthreads();
    if ((int)threadIdx.x < np) {
      float* z = tile + threadIdx.x * CP;
      const int t = load_target(target, p0 + threadIdx.x);
      const bool valid = !(t == ignore_index || t < 0 || t >= C);
      if (valid) {
        float mx = z[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
        const float w = weight ? weight[t] : 1.f;
        if (BWD) {
          const float inv = 1.f / se, wc = w * coef;
          for (int c = 0; c < C; ++c) z[c] = wc * (expf(z[c] - mx) * inv - (c == t ? 1.f : 0.f));
        } else {
          const float nll = (mx + logf(se)) - z[t];
          lsum += (double)(w * nll);
          wsum += (double)w;
        }
      } else if (BWD) {
        for (int c = 0; c < C; ++c) z[c] = 0.f;
      }
    }
    if (BWD) {
      __syncthreads();
      if (ldo == C) {
        float* dst = dlogits + p0 * C;
        for (int e = threadIdx.x; e < np * C; e += 256) dst[e] = tile[(e / C) * CP + (e % C)];
      } else {
        for (int e = threadIdx.x; e < np * C; e += 256) dlogits[(p0 + e / C) * ldo + (e % C)] = tile[(e / C) * CP + (e % C)];
      }
    }
    __syncthreads();
  }
  if (!BWD) {
    __shared__ double red[2][4];
    lsum = wave_sum_d(lsum);
    wsum = wave_sum_d(wsum);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
      red[0][wave] = lsum;
      red[1][wave] = wsum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      partial[2 * blockIdx.x + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
      partial[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
  }
}

// out[0] = loss, out[1] = sum of weights over valid pixels, out[2] = sum of w*nll (raw numerator, for multi-rank normalisation)
__global__ void ce_finalize_kernel(const double* partial, int nblk, float inv_batch, float* out) {
  double l = 0.0, w = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) {
    l += partial[2 * k];
    w += partial[2 * k + 1];
  }
  l = wave_sum_d(l);
  w = wave_sum_d(w);
  if (threadIdx.x == 0) {
    out[0] = (float)(l / w) * inv_batch;
    out[1] = (float)w;
    out[2] = (float)l;
  }
}

// ------------------------------------------------------------------------------------------------ MMD
// X = [gen ; real] (2N rows, D cols).  One wave per 32x32 tile of the 2N x 2N kernel matrix.
struct MmdArgs {
  const float* gen;
  const float* real;
  int ldg, ldr, N, D;
  float* G;        // [2N][2N]: S_ij * sum_v exp(E_ij/v)/v  (for backward)
  double* tile;    // [tiles][2]: (+) and (-) sums
  float sig[8];
  int nsig;
};

__device__ __forceinline__ const float* mmd_row(const MmdArgs& p, int r) {
  return r < p.N ? p.gen + (size_t)r * p.ldg : p.real + (size_t)(r - p.N) * p.ldr;
}

// W waves per tile: wave w, lane-half h own the reduction indices [(2w+h)*seg, +seg), seg = D/(2W); the W partial
// accumulators are added in wave order by wave 0 (deterministic).  W = 4 cuts the dependent load->MFMA chain of this
// latency-bound kernel (64 tiles on 256 CUs) by four.
template <int W>
__global__ __launch_bounds__(64 * W) void mmd_tile_kernel(const MmdArgs p) {
  __shared__ float nrm[2][32];
  __shared__ float nrm_part[W > 1 ? W : 1][2][32];
  __shared__ float acc_all[W > 1 ? W : 1][16][64];
  __shared__ double sum_part[W > 1 ? W : 1][2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int T = (2 * p.N + 31) / 32;
  const int ti = blockIdx.x / T, tj = blockIdx.x % T;
  const int ra = ti * 32 + r, rb = tj * 32 + r;
  const bool va = ra < 2 * p.N, vb = rb < 2 * p.N;
  const float* pa = mmd_row(p, va ? ra : 0);
  const float* pb = mmd_row(p, vb ? rb : 0);
  const int seg = p.D / (2 * W);  // this (wave, lane-half) owns k in [(2w+h)*seg, +seg)
  const int k0 = (2 * w + h) * seg;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  float na = 0.f, nb = 0.f;
  if ((seg & 31) == 0 && ((p.ldg | p.ldr) & 3) == 0) {
    // each lane streams its own row in 128-byte pieces (8 x float4 for A and B), then feeds 32 MFMAs from registers
    // branch-free loads (pa / pb of an out-of-range row point at row 0) and a select afterwards: with the loads under the
    // validity test the compiler waits for each one before issuing the next -- 16 dependent round trips per chunk
    const float* qa = pa + k0;
    const float* qb = pb + k0;
    for (int c = 0; c < seg; c += 32) {
      f32x4 av[8], bv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        av[t] = *reinterpret_cast<const f32x4*>(qa + c + 4 * t);
        bv[t] = *reinterpret_cast<const f32x4*>(qb + c + 4 * t);
      }
      if (!va) {
#pragma unroll
        for (int t = 0; t < 8; ++t) av[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (!vb) {
#pragma unroll
        for (int t = 0; t < 8; ++t) bv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = av[t][e], b = bv[t][e];
          na = fmaf(a, a, na);
          nb = fmaf(b, b, nb);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
  } else {
#pragma unroll 8
    for (int s = 0; s < seg; ++s) {
      const float a = va ? pa[k0 + s] : 0.f;
      const float b = vb ? pb[k0 + s] : 0.f;
      na = fmaf(a, a, na);
      nb = fmaf(b, b, nb);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  na += __shfl_xor(na, 32, 64);
  nb += __shfl_xor(nb, 32, 64);
  // W > 1: every wave stores its partial accumulator; then wave w finishes accumulator registers [QW*w, QW*(w+1)) of the
  // tile (QW = 16 / W): the exp() phase -- 6 bandwidths x 16 entries per lane, the longest part of this latency-bound kernel
  // when one wave did all of it -- is spread over the W waves.  Partial sums are combined in wave order (deterministic).
  constexpr int QW = 16 / W;
  float accq[QW];
  if (W > 1) {
    if (h == 0) {
      nrm_part[w][0][r] = na;
      nrm_part[w][1][r] = nb;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc_all[w][q][lane] = acc[q];
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < QW; ++qq) {
      float t = acc_all[0][QW * w + qq][lane];
#pragma unroll
      for (int ww = 1; ww < W; ++ww) t += acc_all[ww][QW * w + qq][lane];
      accq[qq] = t;
    }
    if (w == 0 && h == 0) {
      float sa = nrm_part[0][0][r], sb = nrm_part[0][1][r];
#pragma unroll
      for (int ww = 1; ww < W; ++ww) {
        sa += nrm_part[ww][0][r];
        sb += nrm_part[ww][1][r];
      }
      nrm[0][r] = sa;
      nrm[1][r] = sb;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) accq[q] = acc[q];
    if (h == 0) {
      nrm[0][r] = na;
      nrm[1][r] = nb;
    }
    __syncthreads();
  }
  const int j = tj * 32 + r;  // column owned by this lane
  const float nj = nrm[1][r];
  const float invN2 = 1.f / ((float)p.N * (float)p.N);
  double pos = 0.0, neg = 0.0;
#pragma unroll
  for (int qq = 0; qq < QW; ++qq) {
    const int q = QW * w + qq;
    const int il = (q & 3) + 8 * (q >> 2) + 4 * h;
    const int i = ti * 32 + il;
    if (i < 2 * p.N && j < 2 * p.N) {
      const float e = accq[qq] - 0.5f * nrm[0][il] - 0.5f * nj;
      float ks = 0.f, kd = 0.f;
      for (int v = 0; v < p.nsig; ++v) {
        const float k = expf(e / p.sig[v]);
        ks += k;
        kd += k / p.sig[v];
      }
      const bool same = (i < p.N) == (j < p.N);
      if (same) pos += (double)ks; else neg += (double)ks;
      if (p.G) p.G[(size_t)i * (2 * p.N) + j] = (same ? kd : -kd) * invN2;
    }
  }
  pos = wave_sum_d(pos);
  neg = wave_sum_d(neg);
  if (W > 1) {
    if (lane == 0) {
      sum_part[w][0] = pos;
      sum_part[w][1] = neg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double tp = sum_part[0][0], tn = sum_part[0][1];
#pragma unroll
      for (int ww = 1; ww < W; ++ww) {
        tp += sum_part[ww][0];
        tn += sum_part[ww][1];
      }
      p.tile[2 * blockIdx.x + 0] = tp;
      p.tile[2 * blockIdx.x + 1] = tn;
    }
  } else if (lane == 0) {
    p.tile[2 * blockIdx.x + 0] = pos;
    p.tile[2 * blockIdx.x + 1] = neg;
  }
}

__global__ void mmd_finalize_kernel(const double* tile, int ntiles, int N, float* loss) {
  if (threadIdx.x == 0) {
    double pos = 0.0, neg = 0.0;  // same summation pattern for both => exact 0 when gen == real
    for (int k = 0; k < ntiles; ++k) {
      pos += tile[2 * k];
      neg += tile[2 * k + 1];
    }
    const double q = (pos - neg) / ((double)N * (double)N);
    loss[0] = sqrtf((float)q);
  }
}

// dgen[k][d] = gout/L * ( sum_j G[k][j] X[j][d] - rowsum_k(G) * gen[k][d] ),  k < N.  W waves per 32x32 tile, each
// owning a slice of the reduction over j (see mmd_tile_kernel).
template <int W>
__global__ __launch_bounds__(64 * W) void mmd_bwd_kernel(const MmdArgs p, const float* loss, const float* gout, float* dgen,
                                                        int ldo, const double* tile_ws, int ntiles, float* loss_ring,
                                                        const long* slot, int ring_len) {
  __shared__ float rs[32];
  __shared__ float rs_part[W > 1 ? W : 1][32];
  __shared__ float acc_part[W > 1 ? W - 1 : 1][16][64];
  __shared__ float loss_sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  if (tile_ws) {   // loss from the tile partial sums (same serial order as mmd_finalize_kernel): saves a launch in the chain
    // The LAST wave does it while the others already stream their operands; its lanes fetch 64 tiles' sums in one burst
    // (a one-thread loop over global memory costs a dependent L2 round trip every few tiles: ~8 us of this kernel's 13),
    // and lane 0 adds them from LDS in the serial order.
    __shared__ double tsum[2][64];
    if (w == W - 1) {
      double pos = 0.0, neg = 0.0;
      for (int base = 0; base < ntiles; base += 64) {
        const int kt = base + lane;
        tsum[0][lane] = kt < ntiles ? tile_ws[2 * kt] : 0.0;
        tsum[1][lane] = kt < ntiles ? tile_ws[2 * kt + 1] : 0.0;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
          const int n = ntiles - base < 64 ? ntiles - base : 64;
          for (int kk = 0; kk < n; ++kk) {
            pos += tsum[0][kk];
            neg += tsum[1][kk];
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (lane == 0) {
        loss_sh = sqrtf((float)((pos - neg) / ((double)p.N * (double)p.N)));
        if (loss_ring && blockIdx.x == 0) {   // the update's loss value for the host (read back once per training step)
          const long sl = slot[0];
          if (sl < ring_len) loss_ring[sl] = loss_sh;
        }
      }
    }
  }
  const int TD = (p.D + 31) / 32;
  const int tk = blockIdx.x / TD, td = blockIdx.x % TD;
  const int k = tk * 32 + r;
  const int d = td * 32 + r;
  const int twoN = 2 * p.N;
  const bool vk = k < p.N, vd = d < p.D;
  const float* grow = p.G + (size_t)(vk ? k : 0) * twoN;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  float rsum = 0.f;
  const int seg = twoN / (2 * W);  // this (wave, lane-half) owns j in [(2w+h)*seg, +seg)
  const int j0 = (2 * w + h) * seg;
  if ((seg & 31) == 0) {
    for (int c = 0; c < seg; c += 32) {
      f32x4 av[8];
      float bv[32];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        av[t] = vk ? *reinterpret_cast<const f32x4*>(grow + j0 + c + 4 * t) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 32; ++t) bv[t] = vd ? mmd_row(p, j0 + c + t)[d] : 0.f;
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        const float a = av[t >> 2][t & 3];
        rsum += a;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc, 0, 0, 0);
      }
    }
  } else {
#pragma unroll 8
    for (int s = 0; s < seg; ++s) {
      const int j = j0 + s;
      const float a = vk ? grow[j] : 0.f;
      const float b = vd ? mmd_row(p, j)[d] : 0.f;
      rsum += a;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  rsum += __shfl_xor(rsum, 32, 64);
  if (h == 0) rs_part[w][r] = rsum;
  if (W > 1 && w > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc_part[w - 1][q][lane] = acc[q];
  }
  __syncthreads();
  if (w > 0) return;
  if (W > 1) {
#pragma unroll
    for (int ww = 1; ww < W; ++ww)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] += acc_part[ww - 1][q][lane];
  }
  if (h == 0) {
    float t = rs_part[0][r];
#pragma unroll
    for (int ww = 1; ww < W; ++ww) t += rs_part[ww][r];
    rs[r] = t;
  }
  __builtin_amdgcn_wave_barrier();
  const float coef = gout[0] / (tile_ws ? loss_sh : loss[0]);
  float gv[16];   // the 16 generated values this lane corrects: one burst of independent loads (clamped, not branched)
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int kk = tk * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    gv[q] = p.gen[(size_t)(kk < p.N ? kk : 0) * p.ldg + (vd ? d : 0)];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int kl = (q & 3) + 8 * (q >> 2) + 4 * h;
    const int kk = tk * 32 + kl;
    if (kk < p.N && vd) dgen[(size_t)kk * ldo + d] = coef * (acc[q] - rs[kl] * gv[q]);
  }
}

}  // namespace

extern "C" int zs3_ce_fwd(const float* logits, int ld, const void* target, int target_is_i64, const float* weight,
                          long P, int C, int ignore_index, int batch, double* partial_ws, float* loss_ws, void* stream) {
  const int nblk = 1024;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)256 * (C + 1) * sizeof(float);
  if (lds > 64 * 1024) return -1;  // C <= 63 (ZS3: 21 and 60 classes)
  if (target_is_i64)
    hipLaunchKernelGGL((ce_tile_kernel<long, false>), dim3(nblk), dim3(256), lds, st, logits, ld, (const long*)target, weight,
                       P, C, ignore_index, partial_ws, nullptr, nullptr, 0.f, nullptr, 0);
  else
    hipLaunchKernelGGL((ce_tile_kernel<float, false>), dim3(nblk), dim3(256), lds, st, logits, ld, (const float*)target,
                       weight, P, C, ignore_index, partial_ws, nullptr, nullptr, 0.f, nullptr, 0);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)partial_ws, nblk,
                     batch > 0 ? 1.f / (float)batch : 1.f, loss_ws);
  return ZS3_LAUNCH_CHECK();
}
extern "C" int zs3_ce_ws_doubles(void) { return 2 * 1024; }

// behind the cross-rank SUM of loss_ws[1..2] (sum of weights, sum of weight * nll): the loss of the gathered batch
__global__ void ce_global_finish_kernel(float* loss_ws, float inv_batch) {
  if (threadIdx.x == 0 && blockIdx.x == 0) loss_ws[0] = loss_ws[2] / loss_ws[1] * inv_batch;
}
extern "C" int zs3_ce_global_finish(float* loss_ws, int global_batch, void* stream) {
  hipLaunchKernelGGL(ce_global_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_ws,
                     global_batch > 0 ? 1.f / (float)global_batch : 1.f);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_ce_bwd(const float* logits, int ld, const void* target, int target_is_i64, const float* weight,
                          long P, int C, int ignore_index, int batch, const float* loss_ws, const float* gout,
                          float* dlogits, int ldo, void* stream) {
  long blocks = (P + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) return 0;
  const float inv_batch = batch > 0 ? 1.f / (float)batch : 1.f;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)256 * (C + 1) * sizeof(float);
  if (lds > 64 * 1024) return -1;  // C <= 63 (ZS3: 21 and 60 classes)
  if (target_is_i64)
    hipLaunchKernelGGL((ce_tile_kernel<long, true>), dim3((int)blocks), dim3(256), lds, st, logits, ld, (const long*)target,
                       weight, P, C, ignore_index, nullptr, loss_ws, gout, inv_batch, dlogits, ldo);
  else
    hipLaunchKernelGGL((ce_tile_kernel<float, true>), dim3((int)blocks), dim3(256), lds, st, logits, ld, (const float*)target,
                       weight, P, C, ignore_index, nullptr, loss_ws, gout, inv_batch, dlogits, ldo);
  return ZS3_LAUNCH_CHECK();
}

static int mmd_fill(MmdArgs& a, const float* gen, int ldg, const float* real, int ldr, int N, int D, const float* sigma,
                    int nsig, float* G, double* tile) {
  if (nsig > 8 || nsig < 1 || D % 2 || N < 1) return -1;
  a.gen = gen; a.real = real; a.ldg = ldg; a.ldr = ldr; a.N = N; a.D = D; a.G = G; a.tile = tile; a.nsig = nsig;
  for (int i = 0; i < nsig; ++i) a.sig[i] = sigma[i];
  return 0;
}

/* sigma: HOST array of nsig bandwidths.  G: [2N][2N] floats (kept for backward), tile_ws: [T*T][2] doubles, T = ceil(2N/32). */
extern "C" int zs3_mmd_fwd(const float* gen, int ldg, const float* real, int ldr, int N, int D, const float* sigma,
                           int nsig, float* G, double* tile_ws, float* loss, void* stream) {
  MmdArgs a;
  if (mmd_fill(a, gen, ldg, real, ldr, N, D, sigma, nsig, G, tile_ws)) return -1;
  const int T = (2 * N + 31) / 32;
  hipStream_t st = (hipStream_t)stream;
  if (D % 256 == 0 && ((ldg | ldr) & 3) == 0)
    hipLaunchKernelGGL(mmd_tile_kernel<4>, dim3(T * T), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(mmd_tile_kernel<1>, dim3(T * T), dim3(64), 0, st, a);
  if (loss) hipLaunchKernelGGL(mmd_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)tile_ws, T * T, N, loss);
  return ZS3_LAUNCH_CHECK();
}

static int mmd_bwd_impl(const float* gen, int ldg, const float* real, int ldr, int N, int D, const float* G,
                        const float* loss, const double* tile_ws, const float* gout, float* dgen, int ldo, void* stream,
                        float* loss_ring = nullptr, const long* slot = nullptr, int ring_len = 0) {
  MmdArgs a;
  float one = 1.f;
  if (mmd_fill(a, gen, ldg, real, ldr, N, D, &one, 1, (float*)G, nullptr)) return -1;
  const int TK = (N + 31) / 32, TD = (D + 31) / 32, T = (2 * N + 31) / 32;
  if (N % 128 == 0)
    hipLaunchKernelGGL(mmd_bwd_kernel<4>, dim3(TK * TD), dim3(256), 0, (hipStream_t)stream, a, loss, gout, dgen, ldo, tile_ws,
                       T * T, loss_ring, slot, ring_len);
  else
    hipLaunchKernelGGL(mmd_bwd_kernel<1>, dim3(TK * TD), dim3(64), 0, (hipStream_t)stream, a, loss, gout, dgen, ldo, tile_ws,
                       T * T, loss_ring, slot, ring_len);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_mmd_bwd(const float* gen, int ldg, const float* real, int ldr, int N, int D, const float* G,
                           const float* loss, const float* gout, float* dgen, int ldo, void* stream) {
  if (!loss) return -1;
  return mmd_bwd_impl(gen, ldg, real, ldr, N, D, G, loss, nullptr, gout, dgen, ldo, stream);
}

extern "C" int zs3_mmd_bwd_ws(const float* gen, int ldg, const float* real, int ldr, int N, int D, const float* G,
                              const double* tile_ws, const float* gout, float* dgen, int ldo, float* loss_ring,
                              const void* slot_dev, int ring_len, void* stream) {
  if (!tile_ws || (loss_ring && !slot_dev)) return -1;
  return mmd_bwd_impl(gen, ldg, real, ldr, N, D, G, nullptr, tile_ws, gout, dgen, ldo, stream, loss_ring, (const long*)slot_dev,
                      ring_len);
}

// End of one captured generator update: the MMD loss value from the tile sums into loss_ring[slot++], Adam step count
// and RNG stream position advanced -- one single-thread launch instead of finalize + two counter bumps + a copy.
__global__ void gmmn_update_epilogue_kernel(const double* tile, int ntiles, int N, float* loss_ring, long* slot, int ring_len,
                                            long* step, long* seed, long seed_inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double pos = 0.0, neg = 0.0;
    for (int k = 0; k < ntiles; ++k) {
      pos += tile[2 * k];
      neg += tile[2 * k + 1];
    }
    const long sl = slot[0];
    if (sl < ring_len) loss_ring[sl] = sqrtf((float)((pos - neg) / ((double)N * (double)N)));
    slot[0] = sl + 1;
    step[0] += 1;
    seed[0] += seed_inc;
  }
}

extern "C" int zs3_gmmn_update_epilogue(const double* tile_ws, int N, float* loss_ring, void* slot_dev, int ring_len,
                                        void* step_dev, void* seed_dev, long seed_inc, void* stream) {
  const int T = (2 * N + 31) / 32;
  hipLaunchKernelGGL(gmmn_update_epilogue_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tile_ws, T * T, N, loss_ring,
                     (long*)slot_dev, ring_len, (long*)step_dev, (long*)seed_dev, seed_inc);
  return ZS3_LAUNCH_CHECK();
}
