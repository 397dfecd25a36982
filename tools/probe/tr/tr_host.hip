// host + device probe of ds_read_b64_tr_b16: prints, for a few address patterns, which LDS element (lds[i] = i) each lane's
// four result values came from.  build: hipcc --offload-arch=gfx950 -O3 tr_host.hip -o tr_host && ./tr_host
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void tr_probe(const int* addr, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int a = addr[threadIdx.x];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
static void run(const char* title, int (*f)(int)) {
  int h[64]; short o[256];
  for (int l = 0; l < 64; ++l) h[l] = f(l);
  int* d; short* dout;
  hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  printf("== %s\n", title);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d addr %5d -> %5d %5d %5d %5d\n", l, h[l], o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  }
  hipFree(d); hipFree(dout);
}
int main() {
  run("linear: lane l reads elements 4l..4l+3", [](int l) { return 4 * l; });
  run("rows of 64 elements: lane l -> row (l%16), column group (l/16)*4", [](int l) { return (l % 16) * 64 + (l / 16) * 4; });
  run("rows of 64 elements: lane l -> row (l%16)/4*... key-major: row = l%4 + 4*(l/16), col = ((l/4)%4)*4", [](int l) { return (l % 4 + 4 * (l / 16)) * 64 + ((l / 4) % 4) * 4; });
  return 0;
}
