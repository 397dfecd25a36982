// Max-pool 3x3/s2 (resnet.py:82,190) and bilinear align_corners=True resize (aspp.py:109,
// decoder.py:34-36, deeplab.py:44,55) on NHWC fp32, forward and backward.  HBM-bound; float4 over
// channels; backward passes are written in gather form (one thread owns an input element), so they
// need no atomics and are deterministic.
#include <cmath>
#include <cstdint>
#include "common.h"
#include "zs3hip.h"

namespace {

template <typename T = float>   // element type of x and out
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* x_, int ldx, float* out_, int ldo,
                                                         unsigned char* idx, int N, int H, int W, int Ho, int Wo, int C,
                                                         int K, int stride, int pad) {
  const T* const x = reinterpret_cast<const T*>(x_);
  T* const out = reinterpret_cast<T*>(out_);
  const int c4n = C >> 2;
  const long total = (long)N * Ho * Wo * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4n) * 4;
    long m = i / c4n;
    const int ow = (int)(m % Wo);
    long r = m / Wo;
    const int oh = (int)(r % Ho), n = (int)(r / Ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    for (int kh = 0; kh < K; ++kh) {
      const int h = oh * stride - pad + kh;
      if (h < 0 || h >= H) continue;
      for (int kw = 0; kw < K; ++kw) {
        const int w = ow * stride - pad + kw;
        if (w < 0 || w >= W) continue;
        f32x4 v = ld4<T>(x + (((long)n * H + h) * W + w) * ldx + cq);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (v[k] > best[k] || v[k] != v[k]) {  // first maximum wins, NaN propagates (ATen semantics)
            best[k] = v[k];
            bi[k] = kh * K + kw;
          }
      }
    }
    st4<T>(out + m * ldo + cq, best);
    if (idx) {
      unsigned pk = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
      *reinterpret_cast<unsigned*>(idx + m * C + cq) = pk;
    }
  }
}

template <typename T = float>   // element type of dy and dx
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* dy_, int ldd, const unsigned char* idx, float* dx_,
                                                         int ldo, int N, int H, int W, int Ho, int Wo, int C, int K,
                                                         int stride, int pad) {
  const T* const dy = reinterpret_cast<const T*>(dy_);
  T* const dx = reinterpret_cast<T*>(dx_);
  const int c4n = C >> 2;
  const long total = (long)N * H * W * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4n) * 4;
    long m = i / c4n;
    const int w = (int)(m % W);
    long r = m / W;
    const int h = (int)(r % H), n = (int)(r / H);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < K; ++kh) {
      const int th = h + pad - kh;
      if (th < 0 || th % stride) continue;
      const int oh = th / stride;
      if (oh >= Ho) continue;
      for (int kw = 0; kw < K; ++kw) {
        const int tw = w + pad - kw;
        if (tw < 0 || tw % stride) continue;
        const int ow = tw / stride;
        if (ow >= Wo) continue;
        const long mo = ((long)n * Ho + oh) * Wo + ow;
        const unsigned pk = *reinterpret_cast<const unsigned*>(idx + mo * C + cq);
        const f32x4 d = ld4<T>(dy + mo * ldd + cq);
        const unsigned tap = (unsigned)(kh * K + kw);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (((pk >> (8 * k)) & 0xFFu) == tap) g[k] += d[k];
      }
    }
    st4<T>(dx + m * ldo + cq, g);
  }
}

__device__ __forceinline__ void src_index(int o, float scale, int in, int& i0, int& i1, float& lam) {
  const float s = scale * (float)o;  // align_corners=True: src = dst * (in-1)/(out-1), all in fp32 as ATen does
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  lam = s - (float)i0;
}

// one bilinear sample; explicit FMA chain so that bilinear_fwd_kernel and the fused upsample+argmax kernel round identically
__device__ __forceinline__ float bilerp(float w00, float w01, float w10, float w11, float a, float b, float c, float d) {
  return fmaf(w11, d, fmaf(w10, c, fmaf(w01, b, w00 * a)));
}

struct ResizeArgs {
  const float* x;
  float* out;
  int N, H, W, Ho, Wo, C, ldx, ldo, accumulate;
  float sh, sw;
};

// One workgroup row (blockIdx.y) per output row (n, oh): the vertical source rows and weights are wave-uniform scalars, and a thread
// finds its (ow, channel group) with ONE 32-bit division.  (Rounds 1-3 flattened everything into one 64-bit index: three 64-bit
// divisions per element made the 21-channel upsample of the class scores -- 88 M elements, the scalar path -- ALU-bound at
// 1.0 TB/s, 346 us per step.)
template <typename T = float>   // element type of x and out
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const ResizeArgs p) {
  T* const pout = reinterpret_cast<T*>(p.out);
  const bool vec = (p.C & 3) == 0 && (p.ldx & 3) == 0 && (p.ldo & 3) == 0;
  const unsigned cn = vec ? (unsigned)(p.C >> 2) : (unsigned)p.C;
  const unsigned per_row = (unsigned)p.Wo * cn;
  for (unsigned row = blockIdx.y; row < (unsigned)(p.N * p.Ho); row += gridDim.y) {
    const unsigned n = row / (unsigned)p.Ho, oh = row - n * (unsigned)p.Ho;
    int h0, h1;
    float lh;
    src_index((int)oh, p.sh, p.H, h0, h1, lh);
    const T* const b0 = reinterpret_cast<const T*>(p.x) + ((long)n * p.H + h0) * p.W * p.ldx;
    const T* const b1 = reinterpret_cast<const T*>(p.x) + ((long)n * p.H + h1) * p.W * p.ldx;
    T* const orow = pout + (long)row * p.Wo * p.ldo;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < per_row; idx += gridDim.x * 256u) {
      const unsigned ow = idx / cn, ci = idx - ow * cn;
      int w0, w1;
      float lw;
      src_index((int)ow, p.sw, p.W, w0, w1, lw);
      const float w00 = (1.f - lh) * (1.f - lw), w01 = (1.f - lh) * lw, w10 = lh * (1.f - lw), w11 = lh * lw;
      if (vec) {
        const int c = (int)ci * 4;
        const f32x4 a00 = ld4<T>(b0 + (long)w0 * p.ldx + c);
        const f32x4 a01 = ld4<T>(b0 + (long)w1 * p.ldx + c);
        const f32x4 a10 = ld4<T>(b1 + (long)w0 * p.ldx + c);
        const f32x4 a11 = ld4<T>(b1 + (long)w1 * p.ldx + c);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = bilerp(w00, w01, w10, w11, a00[e], a01[e], a10[e], a11[e]);
        st4<T>(orow + (long)ow * p.ldo + c, v);
      } else {
        st1<T>(orow + (long)ow * p.ldo + ci, bilerp(w00, w01, w10, w11, ld1<T>(b0 + (long)w0 * p.ldx + ci), ld1<T>(b0 + (long)w1 * p.ldx + ci),
                                                  ld1<T>(b1 + (long)w0 * p.ldx + ci), ld1<T>(b1 + (long)w1 * p.ldx + ci)));
      }
    }
  }
}

// gather-form backward: x = grad wrt the (Ho x Wo) output, out = grad wrt the (H x W) input
__device__ __forceinline__ void cand_range(int i, float scale, int out_n, int& lo, int& hi) {
  if (scale <= 0.f) {
    lo = 0;
    hi = out_n - 1;
    return;
  }
  lo = (int)floorf((float)(i - 1) / scale) - 1;
  hi = (int)ceilf((float)(i + 1) / scale) + 1;
  if (lo < 0) lo = 0;
  if (hi > out_n - 1) hi = out_n - 1;
}

// The candidate window of cand_range is conservative (11 x 11 at the 4x upsample, 7 x 7 of them with a non-zero weight): the
// column weights are computed once per input pixel (not once per candidate row), and rows / columns of zero weight are skipped
// before any address is formed.  BWD_MAXCAND bounds the unrolled column window; larger ratios take the general loop.
constexpr int BWD_MAXCAND = 12;
template <typename T = float>   // element type of the incoming and the produced gradient
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const ResizeArgs p) {
  const bool vec = (p.C & 3) == 0 && (p.ldx & 3) == 0 && (p.ldo & 3) == 0;
  const int cn = vec ? (p.C >> 2) : p.C;
  const long total = (long)p.N * p.H * p.W * cn;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cn);
    long m = i / cn;
    const int w = (int)(m % p.W);
    long r = m / p.W;
    const int h = (int)(r % p.H), n = (int)(r / p.H);
    int olo, ohi, wlo, whi;
    cand_range(h, p.sh, p.Ho, olo, ohi);
    cand_range(w, p.sw, p.Wo, wlo, whi);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const T* g = reinterpret_cast<const T*>(p.x) + (long)n * p.Ho * p.Wo * p.ldx;
    if (whi - wlo < BWD_MAXCAND) {
      float wwv[BWD_MAXCAND];
#pragma unroll
      for (int j = 0; j < BWD_MAXCAND; ++j) {
        const int ow = wlo + j;
        int w0, w1;
        float lw;
        src_index(ow <= whi ? ow : whi, p.sw, p.W, w0, w1, lw);
        wwv[j] = ow <= whi ? (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f) : 0.f;
      }
      for (int oh = olo; oh <= ohi; ++oh) {
        int h0, h1;
        float lh;
        src_index(oh, p.sh, p.H, h0, h1, lh);
        const float wh = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
        if (wh == 0.f) continue;
        // all candidates of the row in one burst of unbranched loads (window columns past `whi` re-read the last one; their
        // weight is zero and a select keeps them out of the sum): a load under `if (weight != 0)` is waited for before the
        // next one is issued
        const T* row = g + ((long)oh * p.Wo + wlo) * p.ldx + (vec ? ci * 4 : ci);
        const int jmax = whi - wlo;
        if (vec) {
          f32x4 v[BWD_MAXCAND];
#pragma unroll
          for (int j = 0; j < BWD_MAXCAND; ++j) v[j] = ld4<T>(row + (long)(j < jmax ? j : jmax) * p.ldx);
#pragma unroll
          for (int j = 0; j < BWD_MAXCAND; ++j)
            if (wwv[j] != 0.f) acc += (wh * wwv[j]) * v[j];
        } else {
          float v[BWD_MAXCAND];
#pragma unroll
          for (int j = 0; j < BWD_MAXCAND; ++j) v[j] = ld1<T>(row + (long)(j < jmax ? j : jmax) * p.ldx);
#pragma unroll
          for (int j = 0; j < BWD_MAXCAND; ++j)
            if (wwv[j] != 0.f) acc[0] += (wh * wwv[j]) * v[j];
        }
      }
    } else {
      for (int oh = olo; oh <= ohi; ++oh) {
        int h0, h1;
        float lh;
        src_index(oh, p.sh, p.H, h0, h1, lh);
        const float wh = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
        if (wh == 0.f) continue;
        for (int ow = wlo; ow <= whi; ++ow) {
          int w0, w1;
          float lw;
          src_index(ow, p.sw, p.W, w0, w1, lw);
          const float ww = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
          if (ww == 0.f) continue;
          const float wt = wh * ww;
          const T* src = g + ((long)oh * p.Wo + ow) * p.ldx;
          if (vec)
            acc += wt * ld4<T>(src + ci * 4);
          else
            acc[0] += wt * ld1<T>(src + ci);
        }
      }
    }
    if (vec) {
      T* dst = reinterpret_cast<T*>(p.out) + m * p.ldo + ci * 4;
      if (p.accumulate) acc += ld4<T>(dst);
      st4<T>(dst, acc);
    } else {
      T* dst = reinterpret_cast<T*>(p.out) + m * p.ldo + ci;
      st1<T>(dst, (p.accumulate ? ld1<T>(dst) : 0.f) + acc[0]);
    }
  }
}

inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int zs3_maxpool_fwd(const float* x, int ldx, float* out, int ldo, void* idx, int N, int H, int W, int Ho,
                               int Wo, int C, int K, int stride, int pad, int io, void* stream) {
  if (C % 4 || ldx % 4 || ldo % 4 || K * K > 255 || (io != 0 && io != 3)) return -1;
  if (io)
    hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t>), dim3(ew_blocks((long)N * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, out, ldo, (unsigned char*)idx, N, H, W, Ho, Wo, C, K, stride, pad);
  else
  hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(ew_blocks((long)N * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     x, ldx, out, ldo, (unsigned char*)idx, N, H, W, Ho, Wo, C, K, stride, pad);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_maxpool_bwd(const float* dy, int ldd, const void* idx, float* dx, int ldo, int N, int H, int W,
                               int Ho, int Wo, int C, int K, int stride, int pad, int io, void* stream) {
  if (C % 4 || ldd % 4 || ldo % 4 || (io != 0 && io != 3)) return -1;
  if (io)
    hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), dim3(ew_blocks((long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       dy, ldd, (const unsigned char*)idx, dx, ldo, N, H, W, Ho, Wo, C, K, stride, pad);
  else
  hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(ew_blocks((long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     dy, ldd, (const unsigned char*)idx, dx, ldo, N, H, W, Ho, Wo, C, K, stride, pad);
  return ZS3_LAUNCH_CHECK();
}

// Validation (train_pascal.py:130-134 + metrics.py:73-82) without the [B,C,H,W] logits ever leaving the GPU -- or, for
// low-resolution logits, ever existing: per output pixel the C class scores are bilinearly sampled from x [N,H,W,C]
// (align_corners=True, the arithmetic of bilinear_fwd_kernel; H == Ho samples exactly), the first maximum is the
// prediction (numpy argmax), and conf[gt*C + pred] is counted for 0 <= gt < C.  Per-block LDS histogram, then integer
// atomics: the result is exact and order-independent.
template <typename T>
__global__ __launch_bounds__(256) void argmax_confusion_kernel(const ResizeArgs p, const T* target,
                                                              unsigned long long* conf) {
  extern __shared__ unsigned hist[];
  const int nbin = p.C * p.C;
  for (int i = threadIdx.x; i < nbin; i += 256) hist[i] = 0u;
  __syncthreads();
  const long total = (long)p.N * p.Ho * p.Wo;
  for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (long)gridDim.x * blockDim.x) {
    const double gtd = (double)target[m];
    if (!(gtd >= 0.0 && gtd < (double)p.C)) continue;
    const int gt = (int)gtd;   // astype(int) truncation of metrics.py:75
    const int ow = (int)(m % p.Wo);
    const long r = m / p.Wo;
    const int oh = (int)(r % p.Ho), n = (int)(r / p.Ho);
    int h0, h1, w0, w1;
    float lh, lw;
    src_index(oh, p.sh, p.H, h0, h1, lh);
    src_index(ow, p.sw, p.W, w0, w1, lw);
    const float* b = p.x + (long)n * p.H * p.W * p.ldx;
    const float* q00 = b + ((long)h0 * p.W + w0) * p.ldx;
    const float* q01 = b + ((long)h0 * p.W + w1) * p.ldx;
    const float* q10 = b + ((long)h1 * p.W + w0) * p.ldx;
    const float* q11 = b + ((long)h1 * p.W + w1) * p.ldx;
    const float w00 = (1.f - lh) * (1.f - lw), w01 = (1.f - lh) * lw, w10 = lh * (1.f - lw), w11 = lh * lw;
    int best = 0;
    float bv = bilerp(w00, w01, w10, w11, q00[0], q01[0], q10[0], q11[0]);
    for (int c = 1; c < p.C; ++c) {
      const float v = bilerp(w00, w01, w10, w11, q00[c], q01[c], q10[c], q11[c]);
      if (v > bv) {
        bv = v;
        best = c;
      }
    }
    atomicAdd(&hist[gt * p.C + best], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nbin; i += 256)
    if (hist[i]) atomicAdd(&conf[i], (unsigned long long)hist[i]);
}

static ResizeArgs make_resize(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int Ho, int Wo, int C,
                              int accumulate) {
  ResizeArgs a;
  a.x = x; a.out = out; a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.C = C; a.ldx = ldx; a.ldo = ldo;
  a.accumulate = accumulate;
  a.sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  a.sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  return a;
}

/* x: [N,H,W,C] -> out: [N,Ho,Wo,C] */
extern "C" int zs3_bilinear_fwd(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int Ho, int Wo,
                                int C, int io, void* stream) {
  if (io != 0 && io != 3) return -1;
  ResizeArgs a = make_resize(x, ldx, out, ldo, N, H, W, Ho, Wo, C, 0);
  if ((long)N * Ho <= 0 || Wo <= 0 || C <= 0) return 0;
  const long per_row = (long)Wo * ((C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0) ? C / 4 : C);
  if (per_row >= (1L << 31) || (long)N * Ho >= (1L << 31)) return -1;
  const long rows = (long)N * Ho;
  const dim3 grid((unsigned)((per_row + 255) / 256 > 64 ? 64 : (per_row + 255) / 256), (unsigned)(rows > 4096 ? 4096 : rows));   // (rows beyond the grid: the kernel's row loop)
  if (io) hipLaunchKernelGGL((bilinear_fwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((bilinear_fwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, a);
  return ZS3_LAUNCH_CHECK();
}

/* dout: [N,Ho,Wo,C] (grad of the resized map) -> dx: [N,H,W,C] */
extern "C" int zs3_bilinear_bwd(const float* dout, int ldd, float* dx, int ldo, int N, int H, int W, int Ho, int Wo,
                                int C, int accumulate, int io, void* stream) {
  if (io != 0 && io != 3) return -1;
  ResizeArgs a = make_resize(dout, ldd, dx, ldo, N, H, W, Ho, Wo, C, accumulate);
  long total = (long)N * H * W * ((C % 4 == 0 && ldd % 4 == 0 && ldo % 4 == 0) ? C / 4 : C);
  if (io) hipLaunchKernelGGL((bilinear_bwd_kernel<bf16_t>), dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((bilinear_bwd_kernel<float>), dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, a);
  return ZS3_LAUNCH_CHECK();
}

/* conf[gt][pred] += 1 over all N*Ho*Wo target pixels with 0 <= gt < C; pred = argmax_c of x [N,H,W,C] bilinearly
 * resized (align_corners=True) to Ho x Wo.  conf: C*C int64 counters (caller zeroes them). */
extern "C" int zs3_argmax_confusion(const float* x, int ldx, int N, int H, int W, int C, const void* target,
                                    int target_is_i64, int Ho, int Wo, void* conf, void* stream) {
  if (C < 1 || C > 128 || N < 1) return -1;
  ResizeArgs a = make_resize(x, ldx, nullptr, 0, N, H, W, Ho, Wo, C, 0);
  const long total = (long)N * Ho * Wo;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const size_t lds = (size_t)C * C * sizeof(unsigned);
  if (target_is_i64)
    hipLaunchKernelGGL(argmax_confusion_kernel<long>, dim3((int)blocks), dim3(256), lds, (hipStream_t)stream, a,
                       (const long*)target, (unsigned long long*)conf);
  else
    hipLaunchKernelGGL(argmax_confusion_kernel<float>, dim3((int)blocks), dim3(256), lds, (hipStream_t)stream, a,
                       (const float*)target, (unsigned long long*)conf);
  return ZS3_LAUNCH_CHECK();
}
