// BatchNorm (+ReLU, +residual) forward/backward around the conv kernels -- HBM-bound passes over NHWC
// fp32 activations, float4-vectorised, deterministic two-level reductions (per-row-chunk partials in
// fp32, cross-chunk combine in fp64).
//
// Forward (train): conv epilogue or zs3_colstats -> partial sums -> zs3_bn_fwd_finalize (mean, invstd,
// fused scale/shift, running-stat update with unbiased variance) -> zs3_affine_act (normalise + residual
// + ReLU, optionally into a channel slice of a concat buffer).
// Backward: zs3_bn_bwd_stats (sum dz, sum dz*xhat) -> zs3_bn_bwd_finalize (dgamma, dbeta, c1, c2) ->
// zs3_bn_act_bwd (dy = gamma*invstd*(dz - c1 - xhat*c2), and dz to the residual branch).
//
// Replaces native_batch_norm fwd/bwd + relu_/threshold_backward + residual add_ at resnet.py:33-53,
// aspp.py:25-29,111-116, decoder.py:30-32,15-24 (113 BN layers; numerics of F.batch_norm, i.e.
// invstd = 1/sqrt(var_biased + eps), running_var uses the unbiased variance).
#include <cstdint>
#include <cstdlib>
// the elementwise kernels of this file read every input element once: non-temporal loads keep them from evicting the operand tiles
// of the convolution / weight-gradient workgroups that run next to them (same-box A/B of the supervised step: 44.81 -> 44.40,
// 44.94 -> 44.41 ms, on a second box 45.79 -> 45.24, 45.64 -> 45.18; non-temporal STORES level or worse -- the next kernel reads
// what these write; the same hint on the conv epilogues' operand loads +0.2 ms, on pooling / resize +0.1, 2-byte mode level;
// tools/probe/r5aa.sh, r5ab.sh)
#ifndef ZS3_NO_LD_NT   // (A/B builds)
#define ZS3_LD_NT 1
#endif
#include "common.h"
#include "zs3hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Column statistics over rows of x[M][ld] (C channels): partial[chunk][2][C].
// MODE 0: (sum x, sum x^2).   MODE 1: BN backward sums (sum dz, sum dz*xhat), dz = relu'(a) * dA.
struct ColArgs {
  const float* x;     // MODE0: x ; MODE1: dA
  const float* a;     // MODE1: post-activation output (ReLU mask) or null
  const float* y;     // MODE1: conv output (pre-BN)
  const float* mean;  // MODE1
  const float* invstd;
  const float* mscale;  // MODE1, optional: ReLU mask recomputed as y*mscale + mshift > 0 (no residual) instead of reading `a`
  const float* mshift;
  const unsigned char* mbits;  // MODE1, optional: ReLU mask as one byte per 4 channels (bit k = channel 4q+k was positive)
  float* partial;
  int M, C, ldx, lda, ldy, rows_per_block;
  float drop_p, drop_inv_keep;          // MODE1, optional: dA is the gradient of dropout(act(bn(y))): its mask is applied first
  unsigned long long drop_seed;
};

template <int MODE, typename T = float>   // T: element type of the activation tensors x / a / y (float or bf16_t)
__global__ __launch_bounds__(256) void colstats_kernel(const ColArgs p) {
  const T* const px = reinterpret_cast<const T*>(p.x);
  const T* const pa = reinterpret_cast<const T*>(p.a);
  const T* const py = reinterpret_cast<const T*>(p.y);
  __shared__ float red[2][256 * 4];
  const int c4n = p.C >> 2;                       // channel quads
  const int tx_n = c4n < 256 ? c4n : 256;         // threads along channels
  const int ty_n = 256 / tx_n;                    // row groups
  const int tid = threadIdx.x;
  const int tx = tid % tx_n, ty = tid / tx_n;
  const int row0 = blockIdx.x * p.rows_per_block;
  const int row1 = min(p.M, row0 + p.rows_per_block);
  for (int cb = 0; cb < c4n; cb += tx_n) {  // uniform trip count: the body contains barriers
    const int cq = cb + tx;
    const bool active = ty < ty_n && cq < c4n;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {0.f, 0.f, 0.f, 0.f};
      f32x4 msc = {0.f, 0.f, 0.f, 0.f}, msh = {0.f, 0.f, 0.f, 0.f};
      if (MODE == 1) {
        mu = *reinterpret_cast<const f32x4*>(p.mean + cq * 4);
        is = *reinterpret_cast<const f32x4*>(p.invstd + cq * 4);
        if (p.mscale) {
          msc = *reinterpret_cast<const f32x4*>(p.mscale + cq * 4);
          msh = *reinterpret_cast<const f32x4*>(p.mshift + cq * 4);
        }
      }
      for (int r = row0 + ty; r < row1; r += ty_n) {
        f32x4 v = ld4<T>(px + (size_t)r * p.ldx + cq * 4);
        if (MODE == 0) {
          s += v;
          q += v * v;
        } else {
          f32x4 yv = ld4<T>(py + (size_t)r * p.ldy + cq * 4);
          if (p.drop_p > 0.f) {
            const unsigned long long e = (unsigned long long)r * p.C + cq * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = u01(p.drop_seed, e + k) >= p.drop_p ? v[k] * p.drop_inv_keep : 0.f;
          }
          if (p.mbits) {
            const unsigned mb = p.mbits[(size_t)r * c4n + cq];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (mb >> k) & 1u ? v[k] : 0.f;
          } else if (p.a) {
            f32x4 av = ld4<T>(pa + (size_t)r * p.lda + cq * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = av[k] > 0.f ? v[k] : 0.f;
          } else if (p.mscale) {
            f32x4 av = yv * msc + msh;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = av[k] > 0.f ? v[k] : 0.f;
          }
          s += v;
          q += v * ((yv - mu) * is);
        }
      }
    }
    // reduce over ty through LDS (fixed order)
    __syncthreads();
    if (active) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        red[0][(ty * tx_n + tx) * 4 + k] = s[k];
        red[1][(ty * tx_n + tx) * 4 + k] = q[k];
      }
    }
    __syncthreads();
    if (active && ty == 0) {
      f32x4 ts = {0.f, 0.f, 0.f, 0.f}, tq = {0.f, 0.f, 0.f, 0.f};
      for (int g = 0; g < ty_n; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ts[k] += red[0][(g * tx_n + tx) * 4 + k];
          tq[k] += red[1][(g * tx_n + tx) * 4 + k];
        }
      float* o0 = p.partial + ((size_t)blockIdx.x * 2 + 0) * p.C + cq * 4;
      float* o1 = p.partial + ((size_t)blockIdx.x * 2 + 1) * p.C + cq * 4;
      *reinterpret_cast<f32x4*>(o0) = ts;
      *reinterpret_cast<f32x4*>(o1) = tq;
    }
  }
}

// FIN_CH channels x FIN_GROUPS row groups per block: thread (ty, tx) sums the partial rows ty, ty + FIN_GROUPS, ... of channel
// c0 + tx in fp64, the row groups are combined through LDS in a fixed order (deterministic; two levels when there are more
// than eight groups); the result is valid in the threads with ty == 0.  These launches sit on the step's dependent chain 226
// times (113 BN layers x forward / backward), each a handful of workgroups whose duration is the number of dependent load
// rounds a thread makes: the shape is chosen for FEW ROWS PER THREAD.  16 x 64 (1024 threads, a quarter-wave reads 64
// contiguous bytes of a partial row) for the usual partial buffers (69-1057 rows: <= 17 rows per thread, two rounds of eight
// loads) and 8 x 128 for the tall ones of the stem and layer1, whose conv kernels write one partial row per 64 output rows
// (4161 / 16513 rows x 64-256 channels).  History, us per call / ms per step: one wave per channel ~10 us; 32 x 8 (round 2;
// 63-132 rows per thread) 6.6 forward / 8.7 backward; 16 x 64 (round 5): -0.22 to -0.33 ms per step on three boxes, same-box
// A/B against 32 x 8 (32 x 32: -0.17, 8 x 128 everywhere: -0.15, 64 x 16: -0.08, 16 x 32: -0.15; tools/probe/r5o.sh).
#ifndef ZS3_FIN_CH   // (tools/probe/build_variant.sh builds the A/B libraries with other shapes)
#define ZS3_FIN_CH 16
#define ZS3_FIN_GROUPS 64
#endif
#ifndef ZS3_FIN_TALL
#define ZS3_FIN_TALL 2048   // partial rows from which the tall form takes over (swept in round 2, re-checked in round 5)
#endif
static int fin_tall_rows() { return ZS3_FIN_TALL; }
template <int FIN_CH, int FIN_GROUPS>
__device__ __forceinline__ bool combine_partials(const float* partial, int chunks, int C, int& c, double& s, double& q) {
  __shared__ double red[2][FIN_GROUPS][FIN_CH];
  const int tx = threadIdx.x & (FIN_CH - 1), ty = threadIdx.x / FIN_CH;
  c = blockIdx.x * FIN_CH + tx;
  if (chunks < 0) {   // `partial` is the SyncBN exchange buffer: fp64 totals [sum x C | sum-of-squares x C | count] (zs3_bn_sync_pack)
    if (ty != 0 || c >= C) return false;
    const double* tot = reinterpret_cast<const double*>(partial);
    s = tot[c];
    q = tot[C + c];
    return true;
  }
  double ls = 0.0, lq = 0.0;
  if (c < C) {
#pragma unroll 8
    for (int k = ty; k < chunks; k += FIN_GROUPS) {
      ls += (double)partial[((size_t)k * 2 + 0) * C + c];
      lq += (double)partial[((size_t)k * 2 + 1) * C + c];
    }
  }
  red[0][ty][tx] = ls;
  red[1][ty][tx] = lq;
  __syncthreads();
  if constexpr (FIN_GROUPS > 8) {   // two levels: eight threads per channel fold FIN_GROUPS / 8 rows each, the owner those eight
    double fs = 0.0, fq = 0.0;
    if (ty < 8) {
#pragma unroll
      for (int i = 0; i < FIN_GROUPS / 8; ++i) {   // groups ty, ty + 8, ...: same order as before, a trip count the compiler can unroll
        fs += red[0][ty + 8 * i][tx];
        fq += red[1][ty + 8 * i][tx];
      }
    }
    __syncthreads();
    if (ty < 8) {
      red[0][ty][tx] = fs;
      red[1][ty][tx] = fq;
    }
    __syncthreads();
  }
  if (ty != 0 || c >= C) return false;
  s = 0.0;
  q = 0.0;
#pragma unroll
  for (int g = 0; g < (FIN_GROUPS > 8 ? 8 : FIN_GROUPS); ++g) {
    s += red[0][g][tx];
    q += red[1][g][tx];
  }
  return true;
}

template <int FIN_CH, int FIN_GROUPS>
__global__ __launch_bounds__(FIN_CH * FIN_GROUPS) void bn_fwd_finalize_kernel(const float* partial, int chunks, int C, double count,
                                                             const double* count_dev, const float* gamma, const float* beta, float eps,
                                                             float momentum, float* running_mean, float* running_var,
                                                             float* mean_out, float* invstd_out, float* scale_out,
                                                             float* shift_out, long* num_batches_tracked, int* range_flag) {
  // range guard, part 1: a flag that is ALREADY up when this launch starts (raised by an earlier layer of this forward pass or by an
  // earlier step the host has not looked at yet; launches are stream-ordered) means this layer's input is what a dead layer
  // produced -- ReLU turns its NaNs into finite zeros, so the sums below look fine and are garbage.  Such a launch leaves every
  // persistent buffer alone: running statistics and num_batches_tracked keep their last good values on every layer downstream
  // of the overflow and in every step until check_forward_range lowers the flag.
  const bool flag_up = range_flag && *reinterpret_cast<volatile int*>(range_flag) != 0;
  if (num_batches_tracked && !flag_up && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
  int c;
  double s, q;
  const bool owner = combine_partials<FIN_CH, FIN_GROUPS>(partial, chunks, C, c, s, q);
  if (count_dev) count = *count_dev;  // cross-rank sample count produced on the device by the SyncBN all-reduce
  if (owner) {
    double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    double invstd = 1.0 / sqrt(var + (double)eps);
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)invstd;
    float sc = g * (float)invstd;
    scale_out[c] = sc;
    shift_out[c] = b - (float)mean * sc;
    // range guard of the f16x3 forward: an operand beyond fp16's range makes the conv output -- and with it
    // these sums -- non-finite.  The layer raises the sticky flag (the fused SGD skips the step while it is up, the trainer falls
    // back to bf16x3 forward products) and leaves the persistent running statistics alone.
    // (the layer that raises the flag itself: its channel groups race on the flag, so its finite channels may or may not update
    // their running statistics and its num_batches_tracked counts the lost step -- one layer, one step; everything behind it is
    // covered by `flag_up`)
    const bool overflow = range_flag && !(isfinite(s) && isfinite(q));
    if (overflow) *range_flag = 1;
    const bool bad = overflow || flag_up;
    if (running_mean && !bad) {
      double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

// SyncBN: this rank's fp64 totals and sample count in the layout the all-reduce carries (one launch instead of a cast, a
// reduction, a fill and two splits on the host side of the exchange)
// (backward pass: this rank's sums ARE its dbeta / dgamma -- written here, the per-rank finalize launch that produced them is gone)
template <int FIN_CH, int FIN_GROUPS>
__global__ __launch_bounds__(FIN_CH * FIN_GROUPS) void bn_sync_pack_kernel(const float* partial, int chunks, int C, double count,
                                                                          double* out, float* dgamma, float* dbeta) {
  int c;
  double s, q;
  const bool owner = combine_partials<FIN_CH, FIN_GROUPS>(partial, chunks, C, c, s, q);
  if (owner) {
    out[c] = s;
    out[C + c] = q;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[2 * C] = count;
}

// eval-mode affine from running statistics
__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int C, float* mean_out, float* invstd_out, float* scale_out, float* shift_out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float invstd = 1.f / sqrtf(rv[c] + eps);
  float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean_out[c] = rm[c];
  invstd_out[c] = invstd;
  scale_out[c] = g * invstd;
  shift_out[c] = b - rm[c] * g * invstd;
}

template <int FIN_CH, int FIN_GROUPS>
__global__ __launch_bounds__(FIN_CH * FIN_GROUPS) void bn_bwd_finalize_kernel(const float* partial, int chunks, int C, double count,
                                                             const double* count_dev, float* dgamma, float* dbeta, float* c1, float* c2,
                                                             int use_batch_stats) {
  int c;
  double s, q;
  const bool owner = combine_partials<FIN_CH, FIN_GROUPS>(partial, chunks, C, c, s, q);
  if (count_dev) count = *count_dev;
  if (owner) {
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
    c1[c] = use_batch_stats ? (float)(s / count) : 0.f;
    c2[c] = use_batch_stats ? (float)(q / count) : 0.f;
  }
}

// out[m][c] = act(alpha * (x[m / div][c] * scale[c] + shift[c]) + res[m][c]) (+ out[m][c] if accumulate)
struct AffArgs {
  const float* x;
  const float* scale;
  const float* shift;
  const float* res;
  float* out;
  unsigned char* mask;   // optional: sign of the pre-activation value, one byte per 4 channels (read back by the BN backward)
  long M;
  int C, ldx, ldr, ldo, div, act, accumulate;
  float alpha, leak;
  float drop_p, drop_inv_keep;          // > 0: out = dropout(act(...)), mask of element (m, c) = u01(drop_seed, m*C + c) >= p
  unsigned long long drop_seed;         // (the mask zs3_dropout draws on the dense [M][C] tensor)
};
// per-channel operands of one float4 column: loaded once per thread when the grid stride is a multiple of the row length (every
// power-of-two channel count: the thread then stays in its column), else once per element
struct AffCol {
  f32x4 scale, shift;
};
__device__ __forceinline__ AffCol aff_col(const AffArgs& p, int cq) {
  AffCol c;
  c.scale = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + cq) : f32x4{1.f, 1.f, 1.f, 1.f};
  c.shift = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + cq) : f32x4{0.f, 0.f, 0.f, 0.f};
  return c;
}
template <typename TI, typename TO>   // element types of x / res (TI) and out (TO)
__device__ __forceinline__ void aff_elem(const AffArgs& p, long i, long m, int cq, const AffCol& col) {
  const long ms = p.div > 1 ? m / p.div : m;
  f32x4 v = ld4<TI>(reinterpret_cast<const TI*>(p.x) + ms * p.ldx + cq);
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
  if (p.res) r = ld4<TI>(reinterpret_cast<const TI*>(p.res) + m * p.ldr + cq);
  if (p.scale) v = v * col.scale;
  if (p.shift) v = v + col.shift;
  v = v * p.alpha;
  if (p.res) v = v + r;
  if (p.mask) p.mask[i] = (unsigned char)((v[0] > 0.f) | ((v[1] > 0.f) << 1) | ((v[2] > 0.f) << 2) | ((v[3] > 0.f) << 3));
  if (p.act == 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
  } else if (p.act == 2) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * p.leak;
  }
  if (p.drop_p > 0.f) {   // nn.Dropout fused behind the activation (aspp.py:100, decoder.py:19,23)
    const unsigned long long e = (unsigned long long)m * p.C + cq;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = u01(p.drop_seed, e + k) >= p.drop_p ? v[k] * p.drop_inv_keep : 0.f;
  }
  TO* dst = reinterpret_cast<TO*>(p.out) + m * p.ldo + cq;
  if (p.accumulate) v = v + ld4<TO>(dst);
  st4<TO>(dst, v);
}
template <typename TI = float, typename TO = float>
__global__ __launch_bounds__(256) void affine_act_kernel(const AffArgs p) {
  const int c4n = p.C >> 2;
  const long total = p.M * c4n;
  const long stride = (long)gridDim.x * blockDim.x;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (stride % c4n == 0) {
    // the thread keeps its channel quad: scale / shift are read once (they were an L1 round trip per element, issued after
    // the data had arrived) and the 64-bit division per element becomes an addition
    long m = i0 / c4n;
    const int cq = (int)(i0 - m * c4n) * 4;
    const long dm = stride / c4n;
    const AffCol col = aff_col(p, cq);
    for (long i = i0; i < total; i += stride, m += dm) aff_elem<TI, TO>(p, i, m, cq, col);
  } else {
    for (long i = i0; i < total; i += stride) {
      const long m = i / c4n;
      const int cq = (int)(i - m * c4n) * 4;
      aff_elem<TI, TO>(p, i, m, cq, aff_col(p, cq));
    }
  }
}

struct BnBwdArgs {
  const float* dA;
  const float* a;
  const float* y;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* c1;
  const float* c2;
  const float* mscale;
  const float* mshift;
  const unsigned char* mbits;
  float* dy;
  float* dres;
  long M;
  int C, ldd, lda, ldy, ldo, ldr, dres_accumulate, act;
  float leak;
  float drop_p, drop_inv_keep;          // > 0: dA is the gradient of the dropout output (see AffArgs)
  unsigned long long drop_seed;
};
struct BwdCol {
  f32x4 is, g, mu, c1, c2, msc, msh;
};
__device__ __forceinline__ BwdCol bwd_col(const BnBwdArgs& p, int cq) {
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  BwdCol c;
  c.is = p.dy ? *reinterpret_cast<const f32x4*>(p.invstd + cq) : one;
  c.g = (p.dy && p.gamma) ? *reinterpret_cast<const f32x4*>(p.gamma + cq) : one;
  c.mu = (p.dy && p.c1) ? *reinterpret_cast<const f32x4*>(p.mean + cq) : zero;
  c.c1 = (p.dy && p.c1) ? *reinterpret_cast<const f32x4*>(p.c1 + cq) : zero;
  c.c2 = (p.dy && p.c1) ? *reinterpret_cast<const f32x4*>(p.c2 + cq) : zero;
  const bool from_y = !p.mbits && !p.a && p.mscale;
  c.msc = from_y ? *reinterpret_cast<const f32x4*>(p.mscale + cq) : one;
  c.msh = from_y ? *reinterpret_cast<const f32x4*>(p.mshift + cq) : zero;
  return c;
}
template <typename T>   // element type of every activation tensor of the call (dA, a, y, dy, dres)
__device__ __forceinline__ void bwd_elem(const BnBwdArgs& p, long i, long m, int cq, const BwdCol& col) {
  f32x4 dz = ld4<T>(reinterpret_cast<const T*>(p.dA) + m * p.ldd + cq);
  f32x4 yv = {0.f, 0.f, 0.f, 0.f};
  if (p.y) yv = ld4<T>(reinterpret_cast<const T*>(p.y) + m * p.ldy + cq);
  if (p.drop_p > 0.f) {   // backward of the fused dropout: the same mask, recomputed
    const unsigned long long e = (unsigned long long)m * p.C + cq;
#pragma unroll
    for (int k = 0; k < 4; ++k) dz[k] = u01(p.drop_seed, e + k) >= p.drop_p ? dz[k] * p.drop_inv_keep : 0.f;
  }
  if (p.mbits) {
    const unsigned mb = p.mbits[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) dz[k] = (mb >> k) & 1u ? dz[k] : (p.act == 2 ? dz[k] * p.leak : 0.f);
  } else if (p.a) {
    f32x4 av = ld4<T>(reinterpret_cast<const T*>(p.a) + m * p.lda + cq);
#pragma unroll
    for (int k = 0; k < 4; ++k) dz[k] = av[k] > 0.f ? dz[k] : (p.act == 2 ? dz[k] * p.leak : 0.f);
  } else if (p.mscale) {
    f32x4 av = yv * col.msc + col.msh;
#pragma unroll
    for (int k = 0; k < 4; ++k) dz[k] = av[k] > 0.f ? dz[k] : 0.f;
  }
  if (p.dres) {
    T* dr = reinterpret_cast<T*>(p.dres) + m * p.ldr + cq;
    f32x4 o = dz;
    if (p.dres_accumulate) o = o + ld4<T>(dr);
    st4<T>(dr, o);
  }
  if (p.dy) {
    f32x4 out;
    if (p.c1) {
      f32x4 xhat = (yv - col.mu) * col.is;
      out = col.g * col.is * (dz - col.c1 - xhat * col.c2);
    } else {
      out = col.g * col.is * dz;
    }
    st4<T>(reinterpret_cast<T*>(p.dy) + m * p.ldo + cq, out);
  }
}
template <typename T = float>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const BnBwdArgs p) {
  const int c4n = p.C >> 2;
  const long total = p.M * c4n;
  const long stride = (long)gridDim.x * blockDim.x;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (stride % c4n == 0) {   // the thread keeps its channel quad: its seven per-channel operands are read once (affine_act_kernel)
    long m = i0 / c4n;
    const int cq = (int)(i0 - m * c4n) * 4;
    const long dm = stride / c4n;
    const BwdCol col = bwd_col(p, cq);
    for (long i = i0; i < total; i += stride, m += dm) bwd_elem<T>(p, i, m, cq, col);
  } else {
    for (long i = i0; i < total; i += stride) {
      const long m = i / c4n;
      const int cq = (int)(i - m * c4n) * 4;
      bwd_elem<T>(p, i, m, cq, bwd_col(p, cq));
    }
  }
}

// out = src[0] + src[1] + ... (fixed order), float4: the gradient of a tensor that feeds several branches (ASPP's input has
// five consumers, layer1's output two) in one pass instead of n-1 pairwise library adds
struct SumArgs {
  const float* src[8];
  float* out;
  long n4;
  int n;
};
template <typename T = float>
__global__ __launch_bounds__(256) void sum_n_kernel(const SumArgs p) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < p.n) v[k] = ld4<T>(reinterpret_cast<const T*>(p.src[k]) + 4 * i);
    f32x4 acc = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k)
      if (k < p.n) acc += v[k];
    st4<T>(reinterpret_cast<T*>(p.out) + 4 * i, acc);
  }
}

// out[g][c] = scale * sum_{r<R} x[(g*R + r)][c]     (global average pool, pooled-branch backward)
// (64 channel quads per block left the ASPP pool -- 16 images x 2048 channels x 1089 pixels, 143 MB -- with 128 workgroups of
// 272-row serial chains: 154 us; 16 quads per block = 512 workgroups, 68 rows per thread.)
constexpr int GCS_TX = 16;
template <typename T = float>   // x and out share the element type
__global__ __launch_bounds__(256) void group_colsum_kernel(const float* x_, int ldx, int R, int C, float scale, float* out_,
                                                          int ldo) {
  const T* const x = reinterpret_cast<const T*>(x_);
  T* const out = reinterpret_cast<T*>(out_);
  __shared__ float red[256 * 4];
  const int c4n = C >> 2;
  const int tx_n = c4n < GCS_TX ? c4n : GCS_TX;   // 16 lanes x 16 B = 256 contiguous bytes of a row; 16 row groups per block
  const int ty_n = 256 / tx_n;
  const int tid = threadIdx.x, tx = tid % tx_n, ty = tid / tx_n;
  const int g = blockIdx.y;
  const int cq = blockIdx.x * tx_n + tx;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const bool ok = ty < ty_n && cq < c4n;
  if (ok) {
#pragma unroll 8
    for (int r = ty; r < R; r += ty_n) s += ld4<T>(x + ((size_t)g * R + r) * ldx + cq * 4);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[tid * 4 + k] = ok ? s[k] : 0.f;
  __syncthreads();
  if (ty == 0 && cq < c4n) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int gg = 0; gg < ty_n; ++gg)
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] += red[((gg * tx_n) + tx) * 4 + k];
    st4<T>(out + (size_t)g * ldo + cq * 4, t * scale);
  }
}

// Grid of the elementwise passes below (affine_act, bn_act_bwd, sum_n): at most 4096 workgroups, every thread walks its channel quad
// down the rows.  Through round 5 the cap was 16384 = one float4 per thread on the step's big tensors -- alone that is the fastest
// form on the largest ones (6.5 TB/s against 6.0), and every thread pays its seven per-channel operand loads for one element.
// Inside the step (same box, interleaved, identical losses): fp32 storage 43.91 / 44.04 ms at 16384, 43.81 / 43.80 at 2048,
// 42.67 at 4096 against 43.85 at 512; 2-byte mode 28.82 / 28.68 at 16384, 27.71 / 27.76 at 8192, **27.43 / 27.37 at 4096**,
// 27.55 / 27.50 at 2048, 27.62 / 27.64 at 1024, 28.94 / 28.90 at 512 (tools/probe/r6z2.sh; ZS3_EW_MAXBLOCKS overrides).
// Four-elements-in-flight, branch-free forms of the two kernels (bit-identical, 20-25 % faster alone on the 17-70 MB tensors, slower
// on the 270 MB ones) bought nothing beyond the cap inside the step and were not kept (docs/LAB_NOTES.md).
inline int ew_blocks(long total) {
  static const long cap = getenv("ZS3_EW_MAXBLOCKS") ? atol(getenv("ZS3_EW_MAXBLOCKS")) : 4096;
  long b = (total + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int zs3_colstats_plan(int M, int C, int* chunks, int* rows_per_block) {
  int c4n = C / 4;
  int tx_n = c4n < 256 ? c4n : 256;
  int ty_n = 256 / (tx_n > 0 ? tx_n : 1);
  int rpb = (M + 511) / 512;   // <= 512 partial rows: the [chunks][2][C] buffer stays small (it is re-read by the finalize kernels)
  int minr = ty_n * 4;
  if (rpb < minr) rpb = minr;
  *rows_per_block = rpb;
  *chunks = (M + rpb - 1) / rpb;
  return 0;
}

extern "C" int zs3_colstats(const float* x, int ldx, int M, int C, float* partial, int io, void* stream) {
  if (C % 4 || ldx % 4 || (io & ~1)) return -1;
  ColArgs a{};
  a.x = x; a.partial = partial; a.M = M; a.C = C; a.ldx = ldx;
  int chunks;
  zs3_colstats_plan(M, C, &chunks, &a.rows_per_block);
  if (io & ZS3_IO_IN16) hipLaunchKernelGGL((colstats_kernel<0, bf16_t>), dim3(chunks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((colstats_kernel<0, float>), dim3(chunks), dim3(256), 0, (hipStream_t)stream, a);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_bn_bwd_stats(const float* dA, int ldd, const float* a_out, int lda, const float* y, int ldy,
                                const float* mean, const float* invstd, const float* mask_scale,
                                const float* mask_shift, const unsigned char* mask_bits, int M, int C, float* partial,
                                float drop_p, unsigned long long drop_seed, int io, void* stream) {
  if (C % 4 || ldd % 4 || ldy % 4 || (a_out && lda % 4) || drop_p < 0.f || drop_p >= 1.f || (io & ~1)) return -1;
  ColArgs a{};
  a.drop_p = drop_p; a.drop_inv_keep = 1.f / (1.f - drop_p); a.drop_seed = drop_seed;
  a.x = dA; a.a = a_out; a.y = y; a.mean = mean; a.invstd = invstd; a.partial = partial;
  a.mscale = mask_scale; a.mshift = mask_shift; a.mbits = mask_bits;
  a.M = M; a.C = C; a.ldx = ldd; a.lda = lda; a.ldy = ldy;
  int chunks;
  zs3_colstats_plan(M, C, &chunks, &a.rows_per_block);
  if (io & ZS3_IO_IN16) hipLaunchKernelGGL((colstats_kernel<1, bf16_t>), dim3(chunks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((colstats_kernel<1, float>), dim3(chunks), dim3(256), 0, (hipStream_t)stream, a);
  return ZS3_LAUNCH_CHECK();
}

static int bn_sync_pack_launch(const float* partial, int chunks, int C, double count, double* totals, float* dgamma, float* dbeta,
                               void* stream) {
  if (chunks <= 0 || C <= 0 || !totals) return -1;
  if (chunks >= fin_tall_rows())
    hipLaunchKernelGGL((bn_sync_pack_kernel<8, 128>), dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)stream, partial, chunks, C,
                       count, totals, dgamma, dbeta);
  else
    hipLaunchKernelGGL((bn_sync_pack_kernel<ZS3_FIN_CH, ZS3_FIN_GROUPS>), dim3((C + ZS3_FIN_CH - 1) / ZS3_FIN_CH), dim3(ZS3_FIN_CH * ZS3_FIN_GROUPS), 0, (hipStream_t)stream, partial, chunks, C,
                       count, totals, dgamma, dbeta);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_bn_sync_pack(const float* partial, int chunks, int C, double count, double* totals, void* stream) {
  return bn_sync_pack_launch(partial, chunks, C, count, totals, nullptr, nullptr, stream);
}

extern "C" int zs3_bn_sync_pack_bwd(const float* partial, int chunks, int C, double count, double* totals, float* dgamma,
                                    float* dbeta, void* stream) {
  return bn_sync_pack_launch(partial, chunks, C, count, totals, dgamma, dbeta, stream);
}

extern "C" int zs3_bn_fwd_finalize(const float* partial, int chunks, int C, double count, const double* count_dev,
                                   const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_var, float* mean_out, float* invstd_out, float* scale_out,
                                   float* shift_out, long* num_batches_tracked, int* range_flag, void* stream) {
  if (chunks >= fin_tall_rows())
    hipLaunchKernelGGL((bn_fwd_finalize_kernel<8, 128>), dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)stream, partial, chunks, C,
                       count, count_dev, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, scale_out,
                       shift_out, num_batches_tracked, range_flag);
  else
    hipLaunchKernelGGL((bn_fwd_finalize_kernel<ZS3_FIN_CH, ZS3_FIN_GROUPS>), dim3((C + ZS3_FIN_CH - 1) / ZS3_FIN_CH), dim3(ZS3_FIN_CH * ZS3_FIN_GROUPS), 0, (hipStream_t)stream, partial, chunks, C,
                       count, count_dev, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, scale_out,
                       shift_out, num_batches_tracked, range_flag);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int C, float* mean_out, float* invstd_out,
                                  float* scale_out, float* shift_out, void* stream) {
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, mean_out, invstd_out, scale_out, shift_out);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_bn_bwd_finalize(const float* partial, int chunks, int C, double count, const double* count_dev,
                                   float* dgamma, float* dbeta,
                                   float* c1, float* c2, int use_batch_stats, void* stream) {
  if (chunks >= fin_tall_rows())
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<8, 128>), dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)stream, partial, chunks, C,
                       count, count_dev, dgamma, dbeta, c1, c2, use_batch_stats);
  else
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<ZS3_FIN_CH, ZS3_FIN_GROUPS>), dim3((C + ZS3_FIN_CH - 1) / ZS3_FIN_CH), dim3(ZS3_FIN_CH * ZS3_FIN_GROUPS), 0, (hipStream_t)stream, partial, chunks, C,
                       count, count_dev, dgamma, dbeta, c1, c2, use_batch_stats);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_affine_act(const float* x, int ldx, const float* scale, const float* shift, float alpha,
                              const float* res, int ldr, float* out, int ldo, long M, int C, int div, int act,
                              float leak, int accumulate, unsigned char* mask_out, float drop_p,
                              unsigned long long drop_seed, int io, void* stream) {
  if (C % 4 || ldx % 4 || ldo % 4 || (res && ldr % 4) || drop_p < 0.f || drop_p >= 1.f || (io & ~3)) return -1;
  if (M <= 0) return 0;
  AffArgs a;
  a.drop_p = drop_p; a.drop_inv_keep = 1.f / (1.f - drop_p); a.drop_seed = drop_seed;
  a.x = x; a.scale = scale; a.shift = shift; a.res = res; a.out = out; a.mask = mask_out; a.M = M; a.C = C; a.ldx = ldx; a.ldr = ldr;
  a.ldo = ldo; a.div = div; a.act = act; a.accumulate = accumulate; a.alpha = alpha; a.leak = leak;
  const dim3 grid(ew_blocks(M * (C / 4)));
  hipStream_t st = (hipStream_t)stream;
  switch (io) {   // bit 0: x / res are bf16, bit 1: out is (a mixed call is the cast between the two storage forms)
    case 0: hipLaunchKernelGGL((affine_act_kernel<float, float>), grid, dim3(256), 0, st, a); break;
    case 1: hipLaunchKernelGGL((affine_act_kernel<bf16_t, float>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((affine_act_kernel<float, bf16_t>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((affine_act_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, a); break;
  }
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_bn_act_bwd(const float* dA, int ldd, const float* a_out, int lda, const float* y, int ldy,
                              const float* mean, const float* invstd, const float* gamma, const float* c1,
                              const float* c2, const float* mask_scale, const float* mask_shift,
                              const unsigned char* mask_bits, float* dy, int ldo, float* dres, int ldr,
                              int dres_accumulate, long M, int C, int act, float leak, float drop_p,
                              unsigned long long drop_seed, int io, void* stream) {
  if (C % 4 || ldd % 4 || (dy && ldo % 4) || (a_out && lda % 4) || (dres && ldr % 4) || drop_p < 0.f || drop_p >= 1.f) return -1;
  if (io != 0 && io != 3) return -1;   // every activation tensor of the call has one element type
  if (M <= 0) return 0;
  BnBwdArgs a;
  a.drop_p = drop_p; a.drop_inv_keep = 1.f / (1.f - drop_p); a.drop_seed = drop_seed;
  a.dA = dA; a.a = a_out; a.y = y; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.c1 = c1; a.c2 = c2;
  a.mscale = mask_scale; a.mshift = mask_shift; a.mbits = mask_bits;
  a.dy = dy; a.dres = dres; a.M = M; a.C = C; a.ldd = ldd; a.lda = lda; a.ldy = ldy; a.ldo = ldo; a.ldr = ldr;
  a.dres_accumulate = dres_accumulate; a.act = act; a.leak = leak;
  if (io) hipLaunchKernelGGL((bn_act_bwd_kernel<bf16_t>), dim3(ew_blocks(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((bn_act_bwd_kernel<float>), dim3(ew_blocks(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, a);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_group_colsum(const float* x, int ldx, int G, int R, int C, float scale, float* out, int ldo, int io,
                                void* stream) {
  if (C % 4 || ldx % 4 || ldo % 4 || (io != 0 && io != 3)) return -1;
  if (G <= 0) return 0;
  int c4n = C / 4;
  int tx_n = c4n < GCS_TX ? c4n : GCS_TX;
  dim3 grid((c4n + tx_n - 1) / tx_n, G);
  if (io) hipLaunchKernelGGL((group_colsum_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, R, C, scale, out, ldo);
  else hipLaunchKernelGGL((group_colsum_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, R, C, scale, out, ldo);
  return ZS3_LAUNCH_CHECK();
}

/* out[i] = src[0][i] + ... + src[n-1][i], 2 <= n <= 8 dense fp32 arrays of `count` elements (count % 4 == 0, 16-byte aligned);
 * srcs: HOST array of n device pointers.  out may alias src[0]. */
extern "C" int zs3_sum_n(const void* const* srcs, int n, float* out, long count, int io, void* stream) {
  if (n < 2 || n > 8 || count <= 0 || (count & 3) || ((uintptr_t)out & 15) || (io != 0 && io != 3)) return -1;
  SumArgs a = {};
  for (int k = 0; k < n; ++k) {
    if (!srcs[k] || ((uintptr_t)srcs[k] & 15)) return -1;
    a.src[k] = (const float*)srcs[k];
  }
  a.out = out; a.n4 = count / 4; a.n = n;
  if (io) hipLaunchKernelGGL((sum_n_kernel<bf16_t>), dim3(ew_blocks(count / 4)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((sum_n_kernel<float>), dim3(ew_blocks(count / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return ZS3_LAUNCH_CHECK();
}
