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
__global__ void prep_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ f_pk,
                                   unsigned short* __restrict__ t_pk, int cout, int taps, int cin, int cin_pad,
                                   int cout_pad) {
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
    unsigned short h = f32_to_bf16_rne(v);
    *dh = h;
    *dl = f32_to_bf16_rne(v - bf16_bits_to_f32(h));
  }
}

// All weights of a model in one launch (after the optimizer step): table[e] = {w, f_pk, t_pk, cout, taps, cin, cin_pad,
// cout_pad}, blockmap[b] = {entry, chunk}; a block converts PREP_CHUNK consecutive elements of the (forward ++ transposed)
// index space of its entry.  Same arithmetic as prep_weight_kernel.
constexpr int PREP_CHUNK = 8192;
__global__ __launch_bounds__(256) void prep_weight_multi_kernel(const long* __restrict__ table,
                                                                const int* __restrict__ blockmap) {
  const int e = blockmap[2 * blockIdx.x], chunk = blockmap[2 * blockIdx.x + 1];
  const long* t = table + 8 * (long)e;
  const float* w = reinterpret_cast<const float*>(t[0]);
  unsigned short* f_pk = reinterpret_cast<unsigned short*>(t[1]);
  unsigned short* t_pk = reinterpret_cast<unsigned short*>(t[2]);
  const int cout = (int)t[3], taps = (int)t[4], cin = (int)t[5], cin_pad = (int)t[6], cout_pad = (int)t[7];
  const long nf = (long)cout * taps * cin_pad;
  const long nt = t_pk ? (long)cin * taps * cout_pad : 0;
  const long i0 = (long)chunk * PREP_CHUNK;
  const long i1 = i0 + PREP_CHUNK < nf + nt ? i0 + PREP_CHUNK : nf + nt;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
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
    unsigned short h = f32_to_bf16_rne(v);
    *dh = h;
    *dl = f32_to_bf16_rne(v - bf16_bits_to_f32(h));
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
                     (unsigned short*)t_pk, cout, taps, cin, cin_pad, cout_pad);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_prep_chunk(void) { return PREP_CHUNK; }

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
