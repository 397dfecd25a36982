// Operand preparation kernels: bf16 hi/lo split of weights (forward and transposed-for-dgrad
// layouts), NCHW -> padded NHWC4 image conversion for the 7x7 stem.
#include "common.h"
#include "zs3hip.h"

namespace {

// w: [cout][taps][cin] fp32 (the channels_last storage of an OIHW parameter).
// forward operand: rows = cout, k = (tap, ci) with ci zero-padded to cin_pad
// dgrad operand  : rows = cin,  k = (tap, co) with co zero-padded to cout_pad
// Both are stored [row][k/32][{hi,lo}][32] bf16: the hi and lo halves of one 32-wide K chunk are adjacent,
// so a (row, chunk) is one 128-byte line for the conv kernel's staging loads.
__device__ __forceinline__ long packed_index(long row, long k, long ktot, int half) {
  return ((row * (ktot >> 5) + (k >> 5)) * 2 + half) * 32 + (k & 31);
}
// fp16 hi / lo of one value (the forward plane of the f16x3 arithmetic, common.h): both halves round to nearest even
__device__ __forceinline__ void split_f16(float v, unsigned short& h, unsigned short& l) {
  v *= ZS3_F16X3_WSCALE;      // (exact; undone on the accumulators of every PREC = 4 kernel: common.h)
  const _Float16 hh = (_Float16)v;
  const _Float16 ll = (_Float16)(v - (float)hh);
  h = __builtin_bit_cast(unsigned short, hh);
  l = __builtin_bit_cast(unsigned short, ll);
}
__global__ void prep_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ f_pk,
                                   unsigned short* __restrict__ t_pk, int cout, int taps, int cin, int cin_pad,
                                   int cout_pad, int f_fmt) {
  const long nf = (long)cout * taps * cin_pad;
  const long nt = t_pk ? (long)cin * taps * cout_pad : 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nt; i += (long)gridDim.x * blockDim.x) {
    float v = 0.f;
    unsigned short *dh, *dl;
    if (i < nf) {
      int ci = (int)(i % cin_pad);
      long r = i / cin_pad;
      int tap = (int)(r % taps), co = (int)(r / taps);
      if (ci < cin) v = w[((long)co * taps + tap) * cin + ci];
      const long ktot = (long)taps * cin_pad, k = (long)tap * cin_pad + ci;
      dh = f_pk + packed_index(co, k, ktot, 0);
      dl = f_pk + packed_index(co, k, ktot, 1);
    } else {
      long kk = i - nf;
      int co = (int)(kk % cout_pad);
      long r = kk / cout_pad;
      int tap = (int)(r % taps), ci = (int)(r / taps);
      if (co < cout) v = w[((long)co * taps + tap) * cin + ci];
      const long ktot = (long)taps * cout_pad, k = (long)tap * cout_pad + co;
      dh = t_pk + packed_index(ci, k, ktot, 0);
      dl = t_pk + packed_index(ci, k, ktot, 1);
    }
    unsigned short h, l;
    if (f_fmt == 1 && i < nf) {
      split_f16(v, h, l);
    } else {
      h = f32_to_bf16_rne(v);
      l = f32_to_bf16_rne(v - bf16_bits_to_f32(h));
    }
    *dh = h;
    *dl = l;
  }
}

// Exact-fp32 test mode (prec = 0, conv_igemm.hip / conv_wgrad.hip on v_mfma_f32_32x32x2_f32): the same two operands as plain
// fp32 [row][k] rows with the same zero padding.  A 32-wide K chunk is 128 bytes here as well, so the plane buffers, their
// strides and the staging loads of conv_igemm_kernel are those of the packed bf16 {hi,lo} form.
__global__ void prep_weight_f32_kernel(const float* __restrict__ w, float* __restrict__ f_pk, float* __restrict__ t_pk, int cout,
                                       int taps, int cin, int cin_pad, int cout_pad) {
  const long nf = (long)cout * taps * cin_pad;
  const long nt = t_pk ? (long)cin * taps * cout_pad : 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nt; i += (long)gridDim.x * blockDim.x) {
    if (i < nf) {
      const int ci = (int)(i % cin_pad);
      const long r = i / cin_pad;
      const int tap = (int)(r % taps), co = (int)(r / taps);
      f_pk[i] = ci < cin ? w[((long)co * taps + tap) * cin + ci] : 0.f;
    } else {
      const long kk = i - nf;
      const int co = (int)(kk % cout_pad);
      const long r = kk / cout_pad;
      const int tap = (int)(r % taps), ci = (int)(r / taps);
      t_pk[kk] = co < cout ? w[((long)co * taps + tap) * cin + ci] : 0.f;
    }
  }
}

// All weights of a model in one launch (after the optimizer step): table[e] = {w, f_pk, t_pk, cout, taps, cin, cin_pad,
// cout_pad}, blockmap[b] = {entry, chunk}.  A chunk is one filter tap x 32 output channels x up to 256 input channels,
// processed as 32x32 tiles: w is read once, coalesced along ci; the forward plane is written from registers (8 bytes
// per lane, rows = co) and the transposed plane through an LDS transpose (rows = ci, 8 bytes per lane along co).  The
// per-element form above reads w with a row stride for the transposed plane: measured 6.7 GB of fabric reads per step.
constexpr int PREP_CI_GROUP = 256;
__global__ __launch_bounds__(256) void prep_weight_multi_kernel(const long* __restrict__ table,
                                                                const int* __restrict__ blockmap) {
  __shared__ unsigned short th[32][36], tl[32][36];   // [ci][co] hi / lo, padded rows
  const int e = blockmap[2 * blockIdx.x];
  int chunk = blockmap[2 * blockIdx.x + 1];
  const long* t = table + 8 * (long)e;
  const float* w = reinterpret_cast<const float*>(t[0]);
  unsigned short* f_pk = reinterpret_cast<unsigned short*>(t[1]);
  unsigned short* t_pk = reinterpret_cast<unsigned short*>(t[2]);
  const int cout = (int)t[3], taps = (int)(t[4] & 0xffffffffL), f_fmt = (int)(t[4] >> 32), cin = (int)t[5], cin_pad = (int)t[6],
            cout_pad = (int)t[7];
  const int ngrp = (cin_pad + PREP_CI_GROUP - 1) / PREP_CI_GROUP, nco = cout_pad / 32;
  const int grp = chunk % ngrp; chunk /= ngrp;
  const int cot = chunk % nco;
  const int tap = chunk / nco;
  const int r = threadIdx.x >> 3, q = threadIdx.x & 7;   // row within the tile, 4-wide column group
  const int co0 = cot * 32;
  const long kf_tot = (long)taps * cin_pad, kt_tot = (long)taps * cout_pad;
  const int ci_end = min(cin_pad, (grp + 1) * PREP_CI_GROUP);
  for (int ci0 = grp * PREP_CI_GROUP; ci0 < ci_end; ci0 += 32) {
    const int co = co0 + r, ci = ci0 + 4 * q;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (co < cout) {
      const float* src = w + ((long)co * taps + tap) * cin + ci;
      if (ci + 3 < cin && (((uintptr_t)src) & 15) == 0) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(src);
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (ci + k < cin) v[k] = src[k];
      }
    }
    unsigned short h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      h[k] = f32_to_bf16_rne(v[k]);
      l[k] = f32_to_bf16_rne(v[k] - bf16_bits_to_f32(h[k]));
    }
    if (co < cout) {   // forward plane: row co, k = tap*cin_pad + ci .. ci+3 (inside one 32-chunk)
      const long k = (long)tap * cin_pad + ci;
      unsigned short* dh = f_pk + packed_index(co, k, kf_tot, 0);
      unsigned short* dl = f_pk + packed_index(co, k, kf_tot, 1);
      unsigned short fh[4], fl[4];
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        if (f_fmt == 1) split_f16(v[k4], fh[k4], fl[k4]);   // fp16 hi / lo: the forward convolutions' f16x3 operand
        else { fh[k4] = h[k4]; fl[k4] = l[k4]; }
      }
      *reinterpret_cast<u32x2*>(dh) = u32x2{(unsigned)fh[0] | ((unsigned)fh[1] << 16), (unsigned)fh[2] | ((unsigned)fh[3] << 16)};
      *reinterpret_cast<u32x2*>(dl) = u32x2{(unsigned)fl[0] | ((unsigned)fl[1] << 16), (unsigned)fl[2] | ((unsigned)fl[3] << 16)};
    }
    if (t_pk) {
      __syncthreads();   // previous tile's readers are done
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        th[4 * q + k][r] = h[k];
        tl[4 * q + k][r] = l[k];
      }
      __syncthreads();
      const int ci_t = ci0 + r;   // transposed plane: row ci, k' = tap*cout_pad + co0 + 4q .. +3
      if (ci_t < cin) {
        const long k = (long)tap * cout_pad + co0 + 4 * q;
        unsigned short* dh = t_pk + packed_index(ci_t, k, kt_tot, 0);
        unsigned short* dl = t_pk + packed_index(ci_t, k, kt_tot, 1);
        *reinterpret_cast<u32x2*>(dh) = u32x2{(unsigned)th[r][4 * q] | ((unsigned)th[r][4 * q + 1] << 16),
                                              (unsigned)th[r][4 * q + 2] | ((unsigned)th[r][4 * q + 3] << 16)};
        *reinterpret_cast<u32x2*>(dl) = u32x2{(unsigned)tl[r][4 * q] | ((unsigned)tl[r][4 * q + 1] << 16),
                                              (unsigned)tl[r][4 * q + 2] | ((unsigned)tl[r][4 * q + 3] << 16)};
      }
    }
  }
}

// img: [N][3][H][W] fp32 (NCHW) -> out: [N][H][Wp][4] with the image at columns [left, left+W) and
// zeros elsewhere (4th channel zero).  Feeds the 7x7/s2 stem as a 7x1 conv over 32-float windows.
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ img, float* __restrict__ out, int N, int H, int W,
                                      int Wp, int left) {
  const long total = (long)N * H * Wp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int wp = (int)(i % Wp);
    long r = i / Wp;
    int h = (int)(r % H), n = (int)(r / H);
    int w = wp - left;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (w >= 0 && w < W) {
      const float* src = img + ((long)n * 3 * H + h) * W + w;
      v[0] = src[0];
      v[1] = src[(long)H * W];
      v[2] = src[2L * H * W];
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
  }
}

}  // namespace

extern "C" int zs3_prep_weight(const float* w, void* f_pk, void* t_pk, int cout, int taps, int cin, int cin_pad,
                               int cout_pad, void* stream) {
  long total = (long)cout * taps * cin_pad + (t_pk ? (long)cin * taps * cout_pad : 0);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) return 0;
  hipLaunchKernelGGL(prep_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)f_pk,
                     (unsigned short*)t_pk, cout, taps, cin, cin_pad, cout_pad, 0);
  return ZS3_LAUNCH_CHECK();
}

// The planes of a layer whose FORWARD runs f16x3 (prec = 4 of zs3_conv_igemm): forward plane fp16 hi / lo, transposed (data-
// gradient) plane bf16 hi / lo as above.
extern "C" int zs3_prep_weight_f16fwd(const float* w, void* f_pk, void* t_pk, int cout, int taps, int cin, int cin_pad,
                                      int cout_pad, void* stream) {
  long total = (long)cout * taps * cin_pad + (t_pk ? (long)cin * taps * cout_pad : 0);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) return 0;
  hipLaunchKernelGGL(prep_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)f_pk,
                     (unsigned short*)t_pk, cout, taps, cin, cin_pad, cout_pad, 1);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_prep_weight_f32(const float* w, void* f_pk, void* t_pk, int cout, int taps, int cin, int cin_pad,
                                   int cout_pad, void* stream) {
  long total = (long)cout * taps * cin_pad + (t_pk ? (long)cin * taps * cout_pad : 0);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) return 0;
  hipLaunchKernelGGL(prep_weight_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)f_pk, (float*)t_pk, cout,
                     taps, cin, cin_pad, cout_pad);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_prep_chunks(int cout_pad, int taps, int cin_pad) {
  return taps * (cout_pad / 32) * ((cin_pad + PREP_CI_GROUP - 1) / PREP_CI_GROUP);
}

extern "C" int zs3_prep_weight_multi(const long* table, const int* blockmap, int nblocks, void* stream) {
  if (nblocks <= 0) return 0;
  hipLaunchKernelGGL(prep_weight_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, table, blockmap);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_nchw3_to_nhwc4(const float* img, float* out, int N, int H, int W, int Wp, int left, void* stream) {
  long total = (long)N * H * Wp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) return 0;
  hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, out, N, H, W, Wp,
                     left);
  return ZS3_LAUNCH_CHECK();
}
