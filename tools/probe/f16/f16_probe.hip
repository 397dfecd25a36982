// Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (the lo halves of an fp16 hi/lo split are subnormal for |x| < 0.12)?
// Does v_cvt_pk_f16_f32 produce them (round-to-nearest-even)?  And is the f16 MFMA issued at the bf16 rate?
// build: hipcc --offload-arch=gfx950 -O3 f16_probe.hip -o f16_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(const float* vals, int n, float* out, unsigned* bits) {
  // each test value v: A[row][k = 0] = fp16(v) for every row, B[k = 0][col] = 1 -> D[row][col] = float(fp16(v)) if nothing flushes
  for (int t = 0; t < n; ++t) {
    unsigned pk;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(vals[t]), "v"(0.f));
    h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (threadIdx.x < 32) {   // k = 0..7 live in lanes 0-31, element 0 = k 0
      a[0] = __builtin_bit_cast(_Float16, (unsigned short)(pk & 0xffff));
      b[0] = (_Float16)1.0f;
    }
    f16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[t] = acc[0]; bits[t] = pk & 0xffff; }
  }
}

template <bool F16>
__global__ void rate(float* sink, int iters) {
  f16v acc[4] = {{0}, {0}, {0}, {0}};
  h8 ah = {1, 1, 1, 1, 1, 1, 1, 1};
  b8 ab = __builtin_bit_cast(b8, ah);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (F16) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ah, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[j], 0, 0, 0);
    }
  }
  float s = 0;
  for (int j = 0; j < 4; ++j) s += acc[j][0];
  if (s == 12345.f) sink[0] = s;
}

int main() {
  const float vals[] = {1.0f, 6.2e-5f, 3.0e-5f, 1.0e-6f, 5.96e-8f, 2.0e-8f, 0.1f + 0.1f * 0.000244f, 70000.f};
  const int n = sizeof(vals) / sizeof(float);
  float *dv, *dout; unsigned* dbits;
  hipMalloc(&dv, sizeof(vals)); hipMalloc(&dout, n * 4); hipMalloc(&dbits, n * 4);
  hipMemcpy(dv, vals, sizeof(vals), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dv, n, dout, dbits);
  float out[16]; unsigned bits[16];
  hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost); hipMemcpy(bits, dbits, n * 4, hipMemcpyDeviceToHost);
  for (int t = 0; t < n; ++t) {
    _Float16 h; unsigned short u = (unsigned short)bits[t]; memcpy(&h, &u, 2);
    printf("v = %.6e  cvt_pk_f16 bits 0x%04x = %.6e  (host RNE %.6e)  through the f16 MFMA: %.6e  %s\n", vals[t], bits[t], (double)(float)h,
           (double)(float)(_Float16)vals[t], out[t], out[t] == (float)h ? "kept" : "CHANGED");
  }
  float* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int pass = 0; pass < 2; ++pass)
    for (int f = 0; f < 2; ++f) {
      hipEventRecord(e0);
      if (f) rate<true><<<1024, 256>>>(sink, iters); else rate<false><<<1024, 256>>>(sink, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = 1024.0 * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
      if (pass) printf("%s MFMA 32x32x16: %.1f TF\n", f ? "f16 " : "bf16", flop / ms * 1e-9);
    }
  return 0;
}
