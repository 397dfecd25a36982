// Are the short forms of the operand splits bit-identical to the ones in csrc/common.h?
//   f16x3 : lo = f16(x - f32(hi))  as  v_fma_mixlo_f16 / v_fma_mixhi_f16 (hi * -1.0 + x, one rounding)   -- 3 VALU per pair instead of 6
//   bf16x3: x - f32(hi)            as  v_dot2_f32_bf16 (hi_pk . {-1, 0} + x)                             -- 4 VALU per pair instead of 6
// hipcc --offload-arch=gfx950 -O3 split_probe.hip -o split_probe && ./split_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

__global__ void k(const float* x, unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  // ---- f16, reference form (common.h: split_pair<4>)
  unsigned hi, lo;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  const f16x2_t h = __builtin_bit_cast(f16x2_t, hi);
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(a - (float)h[0]), "v"(b - (float)h[1]));
  // ---- f16, mixed-precision FMA form
  unsigned lo2;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo2) : "v"(hi), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo2) : "v"(hi), "v"(b));
  // ---- bf16, reference form (split_pair<3>)
  unsigned bh, bl;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(bh) : "v"(a), "v"(b));
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(bl) : "v"(a - __uint_as_float(bh << 16)), "v"(b - __uint_as_float(bh & 0xFFFF0000u)));
  // ---- bf16, dot2 form
  float ra, rb;
  unsigned bl2;
  asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(ra) : "v"(bh), "v"(0x0000BF80u), "v"(a));
  asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(rb) : "v"(bh), "v"(0xBF800000u), "v"(b));
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(bl2) : "v"(ra), "v"(rb));
  out[6 * i + 0] = hi; out[6 * i + 1] = lo; out[6 * i + 2] = lo2;
  out[6 * i + 3] = bh; out[6 * i + 4] = bl; out[6 * i + 5] = bl2;
}

int main() {
  const int n = 1 << 22;
  std::vector<float> x(2 * n);
  srand(7);
  const float specials[] = {0.f, -0.f, 65504.f, 65519.f, 65520.f, 1e5f, -1e5f, 131008.f, 1e-8f, 6e-5f, 6.1e-5f, 5.96e-8f, 2.9e-8f, 1e-40f,
                            INFINITY, -INFINITY, NAN, 3.3e38f, 1.17549435e-38f, 0.1199f, 0.12f, 1.f, 1.0009765625f};
  const int ns = sizeof(specials) / sizeof(float);
  for (int i = 0; i < 2 * n; ++i) {
    if (i < ns * ns * 2) { x[i] = (i & 1) ? specials[(i / 2) % ns] : specials[(i / 2) / ns % ns]; continue; }
    const double u = (rand() + 0.5) / (RAND_MAX + 1.0), v = (rand() + 0.5) / (RAND_MAX + 1.0);
    const double g = sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
    const int e = rand() % 16 - 10;       // magnitudes 1e-10 .. 1e5
    x[i] = (float)(g * pow(10.0, e));
    if (rand() % 64 == 0) { unsigned bits = rand() | ((unsigned)rand() << 16); memcpy(&x[i], &bits, 4); }   // arbitrary bit patterns
  }
  float* dx; unsigned* dout;
  hipMalloc(&dx, 2 * n * 4); hipMalloc(&dout, 6 * (size_t)n * 4);
  hipMemcpy(dx, x.data(), 2 * n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  std::vector<unsigned> o(6 * (size_t)n);
  if (hipMemcpy(o.data(), dout, 6 * (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
  long bad16 = 0, badbf = 0, nan_only16 = 0, nan_onlybf = 0, badbf_finite = 0;
  auto is_nan16 = [](unsigned short h) { return (h & 0x7C00) == 0x7C00 && (h & 0x3FF); };
  auto is_nanbf = [](unsigned short h) { return (h & 0x7F80) == 0x7F80 && (h & 0x7F); };
  for (int i = 0; i < n; ++i) {
    if (o[6 * i + 1] != o[6 * i + 2]) {
      const unsigned a = o[6 * i + 1], b = o[6 * i + 2];
      bool only_nan = true;
      for (int hfl = 0; hfl < 2; ++hfl) { unsigned short p = a >> (16 * hfl), q = b >> (16 * hfl); if (p != q && !(is_nan16(p) && is_nan16(q))) only_nan = false; }
      if (only_nan) ++nan_only16; else if (++bad16 <= 8) printf("f16 mismatch: x = (%g, %g) hi %08x lo ref %08x mix %08x\n", x[2 * i], x[2 * i + 1], o[6 * i], a, b);
    }
    if (o[6 * i + 4] != o[6 * i + 5]) {
      const unsigned a = o[6 * i + 4], b = o[6 * i + 5];
      bool only_nan = true;
      for (int hfl = 0; hfl < 2; ++hfl) { unsigned short p = a >> (16 * hfl), q = b >> (16 * hfl); if (p != q && !(is_nanbf(p) && is_nanbf(q))) only_nan = false; }
      const bool fin = isfinite(x[2 * i]) && isfinite(x[2 * i + 1]) && fabsf(x[2 * i]) < 3e38f && fabsf(x[2 * i + 1]) < 3e38f;
      if (fin && ++badbf_finite <= 16) printf("bf16 FINITE mismatch: x = (%.9g, %.9g) hi %08x lo ref %08x dot2 %08x\n", x[2 * i], x[2 * i + 1], o[6 * i + 3], a, b);
      if (only_nan) ++nan_onlybf; else if (++badbf <= 2) printf("bf16 mismatch: x = (%g, %g) hi %08x lo ref %08x dot2 %08x\n", x[2 * i], x[2 * i + 1], o[6 * i + 3], a, b);
    }
  }
  printf("bf16 dot2 mismatches with both inputs finite (< 3e38): %ld\n", badbf_finite);
  printf("%d pairs: f16 mix form: %ld mismatches (+%ld NaN-payload-only); bf16 dot2 form: %ld mismatches (+%ld NaN-payload-only)\n", n, bad16, nan_only16, badbf, nan_onlybf);
  return 0;
}
