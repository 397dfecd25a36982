// Probe: runtime compat + MFMA operand layouts on gfx950 (scratch tool, not product code).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k_axpy(const float* x, float* y, float a, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i] + y[i];
}
__device__ inline unsigned short f2bf_trunc(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }

// A [32][16] row-major fp32 (exactly bf16-representable), B [16][32] row-major, C [32][32]
__global__ void k_mfma_bf16(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int s = 0; s < 8; ++s) {
    int k = 8 * (l >> 5) + s;
    a[s] = (short)f2bf_trunc(A[(l & 31) * 16 + k]);
    b[s] = (short)f2bf_trunc(B[k * 32 + (l & 31)]);
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    C[row * 32 + (l & 31)] = c[r];
  }
}
// A [32][2], B [2][32]
__global__ void k_mfma_f32(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 31) * 2 + (l >> 5)];
  float b = B[(l >> 5) * 32 + (l & 31)];
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    C[row * 32 + (l & 31)] = c[r];
  }
}
__global__ void k_cvt(const float* x, unsigned* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x[2 * i]), "v"(x[2 * i + 1]));
    out[i] = r;
  }
}
extern "C" {
int probe_axpy(const float* x, float* y, float a, int n, void* stream) {
  hipLaunchKernelGGL(k_axpy, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, a, n);
  return (int)hipGetLastError();
}
int probe_mfma_bf16(const float* A, const float* B, float* C, void* stream) {
  hipLaunchKernelGGL(k_mfma_bf16, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C);
  return (int)hipGetLastError();
}
int probe_mfma_f32(const float* A, const float* B, float* C, void* stream) {
  hipLaunchKernelGGL(k_mfma_f32, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C);
  return (int)hipGetLastError();
}
int probe_cvt(const float* x, unsigned* out, int n, void* stream) {
  hipLaunchKernelGGL(k_cvt, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, n);
  return (int)hipGetLastError();
}
}
