// Small HBM/latency-bound kernels of the GMMN step and the optimisers: counter-based dropout, uniform
// noise, nearest-neighbour down-sampling into pixel rows, row gather / scatter / deterministic
// index-add, fused SGD(momentum, weight decay) and Adam updates.
#include <cstdint>
#include "common.h"
#include "zs3hip.h"

namespace {

inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}

// y[m][c] = keep(seed, m*C + c) ? x[m][c] / (1-p) : 0      (same kernel serves backward with x = dy)
__global__ void dropout_kernel(const float* x, int ldx, float* y, int ldy, long M, int C, float p, float inv_keep,
                               unsigned long long seed, const long* row_idx, const unsigned long long* seed_dev) {
  if (seed_dev) seed += seed_dev[0];   // device-resident stream position (captured launches replay with fresh masks)
  const long total = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C;
    const int c = (int)(i - m * C);
    const float v = x[m * ldx + c];
    const unsigned long long e = row_idx ? (unsigned long long)(row_idx[m] * C + c) : (unsigned long long)i;
    y[m * ldy + c] = u01(seed, e) >= p ? v * inv_keep : 0.f;
  }
}

// float4 form of the above for C % 4 == 0 and 16-byte aligned rows (every dropout of the network: 256 channels): one row
// division per four elements and 16-byte accesses; the mask of element (m, c) is the same u01(seed, m*C + c).  (The scalar
// kernel moved the decoder's 272 MB activations at 2.4 TB/s.)
template <typename T = float>   // element type of x and y
__global__ __launch_bounds__(256) void dropout4_kernel(const float* x_, int ldx, float* y_, int ldy, long M, int C, float p,
                                                       float inv_keep, unsigned long long seed, const long* row_idx,
                                                       const unsigned long long* seed_dev) {
  const T* const x = reinterpret_cast<const T*>(x_);
  T* const y = reinterpret_cast<T*>(y_);
  if (seed_dev) seed += seed_dev[0];
  const int c4 = C >> 2;
  const long total = M * c4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / c4;
    const int c = (int)(i - m * c4) << 2;
    const f32x4 v = ld4<T>(x + m * ldx + c);
    const unsigned long long e = (unsigned long long)((row_idx ? row_idx[m] : m) * C + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = u01(seed, e + k) >= p ? v[k] * inv_keep : 0.f;
    st4<T>(y + m * ldy + c, o);
  }
}

__global__ void uniform_kernel(float* out, long n, unsigned long long seed, const unsigned long long* seed_dev) {
  if (seed_dev) seed += seed_dev[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = u01(seed, (unsigned long long)i);
}

// src: [C][H][W] (one NCHW image) -> rows[(oh*ow)][ldo] with rows[p][c] = src[c][ih(p)][iw(p)], nearest:
// ih = min(floor(oh * (H/ho)), H-1) in fp32 as ATen's upsample_nearest does.  A workgroup transposes a 32-pixel x 32-channel
// tile through LDS: the reads walk a channel plane (consecutive output pixels), the writes walk a pixel row (consecutive
// channels: 128 contiguous bytes per 32 lanes).  (One thread per element wrote 4 bytes at a 1200-byte stride: 85 us per
// 129x129x300 map, 16 maps per GMMN step.)
__global__ __launch_bounds__(256) void nearest_rows_kernel(const float* src, int C, int H, int W, int ho, int wo, float sh,
                                                           float sw, float* rows, int ldo) {
  __shared__ float tile[32][33];
  const int npix = ho * wo;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  const int pix = p0 + lx;
  if (pix < npix) {
    const int oh = pix / wo, ow = pix - oh * wo;
    int ih = (int)floorf((float)oh * sh), iw = (int)floorf((float)ow * sw);
    ih = ih < H - 1 ? ih : H - 1;
    iw = iw < W - 1 ? iw : W - 1;
    const float* sp = src + (long)ih * W + iw;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ly + 8 * k;
      if (c < C) tile[ly + 8 * k][lx] = sp[(long)c * H * W];
    }
  }
  __syncthreads();
  const int c = c0 + lx;
  if (c < C) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = p0 + ly + 8 * k;
      if (p < npix) rows[(long)p * ldo + c] = tile[lx][ly + 8 * k];
    }
  }
}

// out[r][0:Ca] = a[idx[r]][0:Ca]; out[r][Ca:Ca+Cb] = b[r][0:Cb]; out[r][Ca+Cb:ldo] = 0
__global__ void gather_cat_kernel(const float* a, int lda, const long* idx, int Ca, const float* b, int ldb, int Cb,
                                  float* out, int ldo, long n) {
  const long total = n * ldo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / ldo;
    const int c = (int)(i - r * ldo);
    float v = 0.f;
    if (c < Ca) v = a[(idx ? idx[r] : r) * lda + c];
    else if (c < Ca + Cb) v = b[r * ldb + (c - Ca)];
    out[i] = v;
  }
}

// mode 0: out[r][:] = src[idx[r]][:] (gather); mode 1: out[idx[r]][:] = src[r][:] (scatter, idx unique)
__global__ void rows_kernel(const float* src, int lds, const long* idx, float* out, int ldo, long n, int C, int mode) {
  const long total = n * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    if (mode == 0) out[r * ldo + c] = src[idx[r] * lds + c];
    else out[idx[r] * ldo + c] = src[r * lds + c];
  }
}

// out[idx[r]][c] += src[r][c] for r = 0..n-1 in order; thread c owns its column => deterministic with duplicates
__global__ void index_add_rows_kernel(const float* src, int lds, const long* idx, float* out, int ldo, int n, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  for (int r = 0; r < n; ++r) out[idx[r] * ldo + c] += src[(long)r * lds + c];
}

__global__ void sgd_kernel(float* p, const float* g, float* buf, long n, float lr, float momentum, float wd,
                           int nesterov, int first) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float d = g[i] + wd * p[i];
    if (momentum != 0.f) {
      float b = first ? d : momentum * buf[i] + d;
      buf[i] = b;
      d = nesterov ? d + momentum * b : b;
    }
    p[i] -= lr * d;
  }
}

// multi-tensor SGD: one launch for the whole parameter set.  table[e] = {p, g, buf, n, lr|wd (two packed floats), first};
// blockmap[b] = {entry, chunk}: block b updates elements [chunk*CHUNK, (chunk+1)*CHUNK) of entry e.
#define ZS3_SGD_CHUNK 16384
// skip_flag (optional): the sticky range flag of the f16x3 forward (zs3_bn_fwd_finalize).  While it is up the step that produced
// these gradients multiplied operands beyond fp16's range: parameters and momentum stay as they are (a skipped step, like a loss
// scaler's overflow step); a momentum buffer that this step would have created is zeroed so that the next step starts it.
__global__ __launch_bounds__(256) void sgd_multi_kernel(const long* __restrict__ table, const int* __restrict__ blockmap,
                                                       float momentum, int nesterov, const int* __restrict__ skip_flag) {
  const int e = blockmap[2 * blockIdx.x], chunk = blockmap[2 * blockIdx.x + 1];
  const long* t = table + 6L * e;
  float* p = reinterpret_cast<float*>(t[0]);
  const float* g = reinterpret_cast<const float*>(t[1]);
  float* buf = reinterpret_cast<float*>(t[2]);
  const long n = t[3];
  const float lr = __uint_as_float((unsigned)(t[4] & 0xFFFFFFFFL)), wd = __uint_as_float((unsigned)(t[4] >> 32));
  const int first = (int)t[5];
  const long lo = (long)chunk * ZS3_SGD_CHUNK, hi = min(n, lo + ZS3_SGD_CHUNK);
  if (skip_flag && *skip_flag) {
    if (first && momentum != 0.f)
      for (long i = lo + threadIdx.x; i < hi; i += 256) buf[i] = 0.f;
    return;
  }
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    float d = g[i] + wd * p[i];
    if (momentum != 0.f) {
      float b = first ? d : momentum * buf[i] + d;
      buf[i] = b;
      d = nesterov ? d + momentum * b : b;
    }
    p[i] -= lr * d;
  }
}

// torch.optim.Adam (no amsgrad, eps outside sqrt/bias-correction as in torch): step_size = lr / bc1,
// denom = sqrt(v)/sqrt(bc2) + eps
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                            float wd, float bc1, float bc2_sqrt, const long* step_dev) {
  if (step_dev) {  // step count lives on the device (CUDA-graph replays): bias corrections for step_dev[0] + 1
    const double t = (double)(step_dev[0] + 1);
    bc1 = (float)(1.0 - pow((double)b1, t));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] + wd * p[i];
    float mi = m[i] + (1.f - b1) * (gi - m[i]);        // lerp form used by torch
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

// ---- fused pieces of the GMMN generator update (one captured graph per (image, class): every launch counts) ----
// out[r][0:Ca] = a[idx[r]][0:Ca]; out[r][Ca:Ca+Cb] = U[0,1) noise with the stream of uniform_kernel on a [n][Cb]
// tensor (element r*Cb + c); out[r][Ca+Cb:ldo] = 0
__global__ void gather_cat_noise_kernel(const float* a, int lda, const long* idx, int Ca, int Cb, float* out, int ldo,
                                        long n, unsigned long long seed, const unsigned long long* seed_dev,
                                        const long* noise_key) {
  if (seed_dev) seed += seed_dev[0];
  const long total = n * ldo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / ldo;
    const int c = (int)(i - r * ldo);
    float v = 0.f;
    if (c < Ca) v = a[(idx ? idx[r] : r) * lda + c];
    else if (c < Ca + Cb) v = u01(seed, (unsigned long long)((noise_key ? noise_key[r] : r) * Cb + (c - Ca)));
    out[i] = v;
  }
}

// out = dropout_backward(dy) * leaky_relu'(h): the two elementwise steps between the generator's second and first Linear
__global__ void dropout_act_bwd_kernel(const float* dy, int ldd, const float* h, int ldh, float* out, int ldo, long M, int C,
                                       float p, float inv_keep, unsigned long long seed, const long* row_idx,
                                       const unsigned long long* seed_dev, float leak) {
  if (seed_dev) seed += seed_dev[0];
  const long total = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C;
    const int c = (int)(i - m * C);
    float v = dy[m * ldd + c];
    if (p > 0.f) {
      const unsigned long long e = row_idx ? (unsigned long long)(row_idx[m] * C + c) : (unsigned long long)i;
      v = u01(seed, e) >= p ? v * inv_keep : 0.f;
    }
    out[m * ldo + c] = h[m * ldh + c] > 0.f ? v : v * leak;
  }
}

// out[c] = sum_m x[m][c] for the few hundred rows of a bias gradient: 64 channels x 4 interleaved row groups per block,
// the four partial sums combined in a fixed order (deterministic)
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, int ldx, int M, int C, float* out) {
  __shared__ float red[256];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int m = ty; m < M; m += 4) s += x[(long)m * ldx + c];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (ty == 0 && c < C) out[c] = (red[tx] + red[64 + tx]) + (red[128 + tx] + red[192 + tx]);
}

// pix_local[j] = order[ridx[j]], pix_global[j] = pix_local[j] + base: the sampled pixel rows of one (image, class)
__global__ void sample_rows_kernel(const long* order, const long* ridx, long base, long* pix_local, long* pix_global, int s) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < s) {
    const long v = order[ridx[j]];
    pix_local[j] = v;
    pix_global[j] = v + base;
  }
}

// Adam for several tensors in one launch; tensors with plane pointers (Linear weights [cout][cin]) also get their bf16
// hi/lo operands rewritten in place (forward rows = cout, transposed rows = cin; layout of zs3_prep_weight).
// table[e] = {p, g, m, v, n, f_pk, t_pk, cout, cin, cin_pad, cout_pad}; blockmap[b] = {entry, chunk}.
constexpr int ADAM_CHUNK = 1024;   // small chunks: the generator has 0.2 M parameters and the update is latency-bound
__device__ __forceinline__ long packed_index_1tap(long row, long k, long ktot, int half) {
  return ((row * (ktot >> 5) + (k >> 5)) * 2 + half) * 32 + (k & 31);
}
__global__ __launch_bounds__(256) void adam_multi_kernel(const long* __restrict__ table, const int* __restrict__ blockmap,
                                                         float lr, float b1, float b2, float eps, float wd,
                                                         const long* step_dev) {
  const int e = blockmap[2 * blockIdx.x], chunk = blockmap[2 * blockIdx.x + 1];
  const long* t = table + 11 * (long)e;
  float* p = reinterpret_cast<float*>(t[0]);
  const float* g = reinterpret_cast<const float*>(t[1]);
  float* m = reinterpret_cast<float*>(t[2]);
  float* v = reinterpret_cast<float*>(t[3]);
  const long n = t[4];
  unsigned short* f_pk = reinterpret_cast<unsigned short*>(t[5]);
  unsigned short* t_pk = reinterpret_cast<unsigned short*>(t[6]);
  const int cin = (int)t[8], cin_pad = (int)t[9], cout_pad = (int)t[10];
  const double st = (double)(step_dev[0] + 1);
  const float bc1 = (float)(1.0 - pow((double)b1, st));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, st));
  const long i0 = (long)chunk * ADAM_CHUNK;
  const long i1 = i0 + ADAM_CHUNK < n ? i0 + ADAM_CHUNK : n;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float gi = g[i] + wd * p[i];
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float w = p[i] - (lr / bc1) * (mi / denom);
    p[i] = w;
    if (f_pk) {
      const long co = i / cin, ci = i - co * cin;
      const unsigned short h = f32_to_bf16_rne(w);
      const unsigned short l = f32_to_bf16_rne(w - bf16_bits_to_f32(h));
      f_pk[packed_index_1tap(co, ci, cin_pad, 0)] = h;
      f_pk[packed_index_1tap(co, ci, cin_pad, 1)] = l;
      if (t_pk) {
        t_pk[packed_index_1tap(ci, co, cout_pad, 0)] = h;
        t_pk[packed_index_1tap(ci, co, cout_pad, 1)] = l;
      }
    }
  }
}

__global__ void counter_add2_kernel(long* c0, long v0, long* c1, long v1) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    c0[0] += v0;
    if (c1) c1[0] += v1;
  }
}


// ------------------------------------------------------------------------------------------- label maps of the GMMN step
// One workgroup per image: the label map at feature resolution (F.interpolate(mode="nearest") of train_pascal_GMMN.py:175-180,
// the index arithmetic of nearest_rows_kernel), its class histogram, and the pixels grouped by class in ascending pixel order
// (= a stable argsort of the labels; a counting sort, since labels are bytes).  Replaces ~20 small library launches per step
// (float cast, transpose, scatter_add histogram, a 12-kernel segmented merge sort, where/zeros: 1.2 ms per B=16 step).
// Pass 1: labels + LDS histogram.  Pass 2, 1024 pixels at a time: a lane's rank among the equal labels of its wave comes from
// ballots, the wave's count per label goes through LDS, position = class base + counts of earlier waves + rank; the class
// bases then advance by the chunk's counts.  Integer arithmetic only: exact and deterministic.
template <typename TT>
__global__ __launch_bounds__(1024) void label_order_kernel(const TT* target, int H, int W, int ho, int wo, float sh, float sw,
                                                           long* tgt_l, long* tgt_cls, long* hist, long* order) {
  __shared__ int cnt[256];
  __shared__ int base[256];
  __shared__ unsigned short wave_cnt[16][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int npix = ho * wo;
  const TT* src = target + (long)blockIdx.x * H * W;
  long* lab_out = tgt_l + (long)blockIdx.x * npix;
  long* cls_out = tgt_cls + (long)blockIdx.x * npix;
  long* ord_out = order + (long)blockIdx.x * npix;
  if (tid < 256) cnt[tid] = 0;
  __syncthreads();
  for (int p = tid; p < npix; p += 1024) {
    const int oh = p / wo, ow = p - oh * wo;
    int ih = (int)floorf((float)oh * sh), iw = (int)floorf((float)ow * sw);
    ih = ih < H - 1 ? ih : H - 1;
    iw = iw < W - 1 ? iw : W - 1;
    long v = (long)src[(long)ih * W + iw];
    if (v < 0 || v > 255) v = 255;            // not a class id: treated like the ignore label
    lab_out[p] = v;
    cls_out[p] = v == 255 ? 0 : v;            // the dataloader's embedding lookup maps 255 to class 0 (base.py:47-48)
    atomicAdd(&cnt[(int)v], 1);
  }
  __syncthreads();
  if (tid < 256) hist[(long)blockIdx.x * 256 + tid] = cnt[tid];
  if (tid == 0) {
    int run = 0;
    for (int c = 0; c < 256; ++c) {
      base[c] = run;
      run += cnt[c];
    }
  }
  __syncthreads();
  for (int c0 = 0; c0 < npix; c0 += 1024) {
    for (int e = tid; e < 16 * 256; e += 1024) (&wave_cnt[0][0])[e] = 0;
    __syncthreads();
    const int p = c0 + tid;
    const bool valid = p < npix;
    const int lab = valid ? (int)lab_out[p] : -1;
    int rank = 0;
    bool todo = valid;
    unsigned long long pending = __ballot(todo);
    while (pending) {
      const int first = __ffsll((long long)pending) - 1;
      const int c = __shfl(lab, first, 64);
      const unsigned long long same = __ballot(todo && lab == c);
      if (todo && lab == c) {
        rank = __popcll(same & ((1ull << lane) - 1ull));
        todo = false;
      }
      if (lane == first) wave_cnt[wave][c] = (unsigned short)__popcll(same);
      pending &= ~same;
    }
    __syncthreads();
    if (valid) {
      int off = base[lab] + rank;
      for (int w = 0; w < wave; ++w) off += wave_cnt[w][lab];
      ord_out[off] = p;
    }
    __syncthreads();
    if (tid < 256) {
      int add = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) add += wave_cnt[w][tid];
      base[tid] += add;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int zs3_dropout(const float* x, int ldx, float* y, int ldy, long M, int C, float p, unsigned long long seed,
                           const long* row_idx, const void* seed_dev, int io, void* stream) {
  if (M <= 0) return 0;
  if (io != 0 && io != 3) return -1;
  if (io) {   // bf16 storage: the vector form only (every dropout of the network has 256 channels)
    if (C % 4 || ldx % 4 || ldy % 4 || (((uintptr_t)x | (uintptr_t)y) & 7)) return -1;
    hipLaunchKernelGGL((dropout4_kernel<bf16_t>), dim3(ew_blocks(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, M,
                       C, p, 1.f / (1.f - p), seed, row_idx, (const unsigned long long*)seed_dev);
  } else if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
    hipLaunchKernelGGL((dropout4_kernel<float>), dim3(ew_blocks(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, M, C,
                       p, 1.f / (1.f - p), seed, row_idx, (const unsigned long long*)seed_dev);
  else
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_blocks(M * C)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, M, C, p,
                       1.f / (1.f - p), seed, row_idx, (const unsigned long long*)seed_dev);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_uniform(float* out, long n, unsigned long long seed, const void* seed_dev, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(uniform_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, out, n, seed,
                     (const unsigned long long*)seed_dev);
  return ZS3_LAUNCH_CHECK();
}

/* target: [B][H][W] float32 or int64 label maps -> tgt_l [B][ho*wo] (nearest resize, values outside 0..255 become 255),
 * tgt_cls (255 -> 0), hist [B][256], order [B][ho*wo] = stable argsort of tgt_l along the pixels; all int64. */
extern "C" int zs3_label_order(const void* target, int target_is_i64, int B, int H, int W, int ho, int wo, long* tgt_l,
                               long* tgt_cls, long* hist, long* order, void* stream) {
  if (B < 1 || ho < 1 || wo < 1) return -1;
  const float sh = (float)H / (float)ho, sw = (float)W / (float)wo;
  if (target_is_i64)
    hipLaunchKernelGGL(label_order_kernel<long>, dim3(B), dim3(1024), 0, (hipStream_t)stream, (const long*)target, H, W, ho, wo,
                       sh, sw, tgt_l, tgt_cls, hist, order);
  else
    hipLaunchKernelGGL(label_order_kernel<float>, dim3(B), dim3(1024), 0, (hipStream_t)stream, (const float*)target, H, W, ho,
                       wo, sh, sw, tgt_l, tgt_cls, hist, order);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_nearest_rows(const float* src, int C, int H, int W, int ho, int wo, float* rows, int ldo,
                                void* stream) {
  const float sh = (float)H / (float)ho, sw = (float)W / (float)wo;
  hipLaunchKernelGGL(nearest_rows_kernel, dim3((ho * wo + 31) / 32, (C + 31) / 32), dim3(256), 0, (hipStream_t)stream, src, C,
                     H, W, ho, wo, sh, sw, rows, ldo);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_gather_cat(const float* a, int lda, const long* idx, int Ca, const float* b, int ldb, int Cb,
                              float* out, int ldo, long n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gather_cat_kernel, dim3(ew_blocks(n * ldo)), dim3(256), 0, (hipStream_t)stream, a, lda, idx, Ca, b,
                     ldb, Cb, out, ldo, n);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_gather_rows(const float* src, int lds, const long* idx, float* out, int ldo, long n, int C,
                               void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(rows_kernel, dim3(ew_blocks(n * C)), dim3(256), 0, (hipStream_t)stream, src, lds, idx, out, ldo, n,
                     C, 0);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_scatter_rows(const float* src, int lds, const long* idx, float* out, int ldo, long n, int C,
                                void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(rows_kernel, dim3(ew_blocks(n * C)), dim3(256), 0, (hipStream_t)stream, src, lds, idx, out, ldo, n,
                     C, 1);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_index_add_rows(const float* src, int lds, const long* idx, float* out, int ldo, int n, int C,
                                  void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(index_add_rows_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, lds, idx, out,
                     ldo, n, C);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_sgd_step(float* p, const float* g, float* buf, long n, float lr, float momentum, float wd,
                            int nesterov, int first, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sgd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr, momentum, wd,
                     nesterov, first);
  return ZS3_LAUNCH_CHECK();
}

// dst[m][0:C] = src[m][0:C], dst[m][C:ldd] = 0 (T = 4- or 2-byte elements, moved as raw bits)
template <typename T>
__global__ void pad_rows_kernel(const T* __restrict__ src, int lds, int C, T* __restrict__ dst, int ldd, long M) {
  const long total = M * ldd;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / ldd;
    const int c = (int)(i - m * ldd);
    dst[i] = c < C ? src[m * lds + c] : T(0);
  }
}

// pack: dst[r][j][k] (G x C per row) = src[r][j][k] (g x c per row) for j < g, k < c, else 0;  unpack: the inverse selection
__global__ void repack_pad_kernel(const float* __restrict__ src, long rows, int g, int c, float* __restrict__ dst, int G, int C,
                                  int unpack) {
  const int per = unpack ? g * c : G * C;
  const long total = rows * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / per;
    const int e = (int)(i - r * per);
    if (unpack) {
      const int j = e / c, k = e - j * c;
      dst[i] = src[r * (G * C) + j * C + k];
    } else {
      const int j = e / C, k = e - j * C;
      dst[i] = (j < g && k < c) ? src[r * (g * c) + j * c + k] : 0.f;
    }
  }
}

extern "C" int zs3_repack_pad(const float* src, long rows, int g, int c, float* dst, int G, int C, int unpack, void* stream) {
  if (rows <= 0) return 0;
  if (g > G || c > C || g < 1 || c < 1) return -1;
  hipLaunchKernelGGL(repack_pad_kernel, dim3(ew_blocks(rows * (unpack ? g * c : G * C))), dim3(256), 0, (hipStream_t)stream, src,
                     rows, g, c, dst, G, C, unpack);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_fill_zero(void* dst, long bytes, void* stream) {
  if (bytes <= 0) return 0;
  return (int)hipMemsetAsync(dst, 0, (size_t)bytes, (hipStream_t)stream);
}

extern "C" int zs3_pad_rows(const void* src, int lds, int C, void* dst, int ldd, long M, int io, void* stream) {
  if (M <= 0) return 0;
  if (C > ldd || (io != 0 && io != 3)) return -1;
  if (io == 3)
    hipLaunchKernelGGL(pad_rows_kernel<unsigned short>, dim3(ew_blocks(M * ldd)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)src, lds, C, (unsigned short*)dst, ldd, M);
  else
    hipLaunchKernelGGL(pad_rows_kernel<unsigned>, dim3(ew_blocks(M * ldd)), dim3(256), 0, (hipStream_t)stream, (const unsigned*)src,
                       lds, C, (unsigned*)dst, ldd, M);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_sgd_chunk(void) { return ZS3_SGD_CHUNK; }
extern "C" int zs3_sgd_multi(const void* table, const void* blockmap, int nblocks, float momentum, int nesterov,
                             const int* skip_flag, void* stream) {
  if (nblocks <= 0) return 0;
  hipLaunchKernelGGL(sgd_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const long*)table,
                     (const int*)blockmap, momentum, nesterov, skip_flag);
  return ZS3_LAUNCH_CHECK();
}

// zs3_sgd_multi with the learning rate / weight decay of the parameter GROUPS as launch arguments: table[e][4] is the entry's
// group index and {lr, wd} of up to ZS3_SGD_MAX_GROUPS groups travel by value with the launch.  The table then depends on nothing
// that changes from step to step (a poly schedule changes lr every iteration): a recorded plan patches the launch argument, an
// eager step needs no per-step host-to-device copy for the schedule.
#define ZS3_SGD_MAX_GROUPS 8
struct SgdGroups {
  float lr[ZS3_SGD_MAX_GROUPS], wd[ZS3_SGD_MAX_GROUPS];
};
__global__ __launch_bounds__(256) void sgd_multi_g_kernel(const long* __restrict__ table, const int* __restrict__ blockmap,
                                                         float momentum, int nesterov, const int* __restrict__ skip_flag,
                                                         const SgdGroups groups) {
  const int e = blockmap[2 * blockIdx.x], chunk = blockmap[2 * blockIdx.x + 1];
  const long* t = table + 6L * e;
  float* p = reinterpret_cast<float*>(t[0]);
  const float* g = reinterpret_cast<const float*>(t[1]);
  float* buf = reinterpret_cast<float*>(t[2]);
  const long n = t[3];
  const int grp = (int)t[4] & (ZS3_SGD_MAX_GROUPS - 1);
  // (selected with compares: a dynamic index into a by-value kernel argument would send the struct through scratch memory)
  float lr = groups.lr[0], wd = groups.wd[0];
#pragma unroll
  for (int k = 1; k < ZS3_SGD_MAX_GROUPS; ++k)
    if (grp == k) {
      lr = groups.lr[k];
      wd = groups.wd[k];
    }
  const int first = (int)t[5];
  const long lo = (long)chunk * ZS3_SGD_CHUNK, hi = min(n, lo + ZS3_SGD_CHUNK);
  if (skip_flag && *skip_flag) {
    if (first && momentum != 0.f)
      for (long i = lo + threadIdx.x; i < hi; i += 256) buf[i] = 0.f;
    return;
  }
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    float d = g[i] + wd * p[i];
    if (momentum != 0.f) {
      float b = first ? d : momentum * buf[i] + d;
      buf[i] = b;
      d = nesterov ? d + momentum * b : b;
    }
    p[i] -= lr * d;
  }
}

extern "C" int zs3_sgd_max_groups(void) { return ZS3_SGD_MAX_GROUPS; }
extern "C" int zs3_sgd_multi_g(const void* table, const void* blockmap, int nblocks, float momentum, int nesterov,
                               const int* skip_flag, const float* group_lr_wd, int ngroups, void* stream) {
  if (nblocks <= 0) return 0;
  if (ngroups < 1 || ngroups > ZS3_SGD_MAX_GROUPS || !group_lr_wd) return -3;
  SgdGroups groups;
  for (int k = 0; k < ZS3_SGD_MAX_GROUPS; ++k) {
    groups.lr[k] = k < ngroups ? group_lr_wd[2 * k] : 0.f;
    groups.wd[k] = k < ngroups ? group_lr_wd[2 * k + 1] : 0.f;
  }
  hipLaunchKernelGGL(sgd_multi_g_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const long*)table,
                     (const int*)blockmap, momentum, nesterov, skip_flag, groups);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                             float eps, float wd, int step, const void* step_dev, void* stream) {
  if (n <= 0) return 0;
  const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2, eps,
                     wd, (float)bc1, (float)sqrt(bc2), (const long*)step_dev);
  return ZS3_LAUNCH_CHECK();
}

__global__ void counter_add_kernel(long* c, long v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += v;
}
extern "C" int zs3_counter_add(void* counter, long v, void* stream) {
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long*)counter, v);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_gather_cat_noise(const float* a, int lda, const long* idx, int Ca, int Cb, float* out, int ldo, long n,
                                    unsigned long long seed, const void* seed_dev, const long* noise_key, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gather_cat_noise_kernel, dim3(ew_blocks(n * ldo)), dim3(256), 0, (hipStream_t)stream, a, lda, idx, Ca,
                     Cb, out, ldo, n, seed, (const unsigned long long*)seed_dev, noise_key);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_dropout_act_bwd(const float* dy, int ldd, const float* h, int ldh, float* out, int ldo, long M, int C,
                                   float p, unsigned long long seed, const long* row_idx, const void* seed_dev, float leak,
                                   void* stream) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(dropout_act_bwd_kernel, dim3(ew_blocks(M * C)), dim3(256), 0, (hipStream_t)stream, dy, ldd, h, ldh,
                     out, ldo, M, C, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed, row_idx,
                     (const unsigned long long*)seed_dev, leak);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_colsum(const float* x, int ldx, int M, int C, float* out, void* stream) {
  if (C <= 0) return 0;
  hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, M, C, out);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_adam_chunk(void) { return ADAM_CHUNK; }
extern "C" int zs3_adam_multi(const long* table, const int* blockmap, int nblocks, float lr, float b1, float b2, float eps,
                              float wd, const void* step_dev, void* stream) {
  if (nblocks <= 0) return 0;
  if (!step_dev) return -1;
  hipLaunchKernelGGL(adam_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, table, blockmap, lr, b1, b2, eps,
                     wd, (const long*)step_dev);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_counter_add2(void* c0, long v0, void* c1, long v1, void* stream) {
  hipLaunchKernelGGL(counter_add2_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long*)c0, v0, (long*)c1, v1);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_sample_rows(const long* order, const long* ridx, long base, long* pix_local, long* pix_global, int s,
                               void* stream) {
  if (s <= 0) return 0;
  hipLaunchKernelGGL(sample_rows_kernel, dim3((s + 255) / 256), dim3(256), 0, (hipStream_t)stream, order, ridx, base,
                     pix_local, pix_global, s);
  return ZS3_LAUNCH_CHECK();
}
