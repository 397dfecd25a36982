// What does "the last workgroup finalizes" cost on MI355X (8 XCDs, one L2 each)?  A stand-in for a conv epilogue: every workgroup
// stores a 128 x 128 fp32 tile and a row of 2 x 128 partial sums; then either (A) exits and a second tiny kernel combines the partial
// rows (what bn_fwd_finalize does today), or (B) fences, takes a ticket, and the last workgroup combines them itself.
// build: hipcc --offload-arch=gfx950 -O3 ticket_probe.hip -o ticket_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void tile_kernel(float* out, float* partial, unsigned* ticket, float* result, int mode, int C) {
  const int tid = threadIdx.x;
  float* t = out + (size_t)blockIdx.x * 128 * 128;
  float acc = 0.f;
  for (int i = tid; i < 128 * 128; i += 256) {
    const float v = (float)((blockIdx.x * 31 + i) & 255) * (1.f / 256.f);
    t[i] = v;
    acc += v;
  }
  if (tid < C) {
    if (mode >= 3) {   // the partial rows as agent-scope (L2-bypassing) stores: no cache write-back needed to publish them
      __hip_atomic_store(&partial[((size_t)blockIdx.x * 2 + 0) * C + tid], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&partial[((size_t)blockIdx.x * 2 + 1) * C + tid], acc * acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      partial[((size_t)blockIdx.x * 2 + 0) * C + tid] = acc;
      partial[((size_t)blockIdx.x * 2 + 1) * C + tid] = acc * acc;
    }
  }
  if (mode == 0) return;
  __shared__ int last;
  if (mode >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores above have been acknowledged
  else __threadfence();
  __syncthreads();
  if (tid == 0) {
    const unsigned n = mode >= 3 ? __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = n == gridDim.x - 1;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last || mode == 1 || mode == 4) return;          // modes 1 / 4: publish + ticket only (what every workgroup pays)
  if (mode < 3) __threadfence();
  // the last workgroup combines: 32 columns x 8 row groups per pass, eight independent loads in flight per thread
  __shared__ double ss[8][32], qq[8][32];
  const int lc = tid & 31, g = tid >> 5;
  for (int c0 = 0; c0 < C; c0 += 32) {
    const int c = c0 + lc;
    double s = 0.0, q = 0.0;
    for (unsigned b0 = g; b0 < gridDim.x; b0 += 64) {
      float v[8], w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned b = b0 + 8 * k;
        const bool ok = b < gridDim.x && c < C;
        const size_t o = ok ? ((size_t)b * 2 + 0) * C + c : 0;
        v[k] = mode >= 3 ? __hip_atomic_load(&partial[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : partial[o];
        w[k] = mode >= 3 ? __hip_atomic_load(&partial[o + C], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : partial[o + C];
        if (!ok) { v[k] = 0.f; w[k] = 0.f; }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { s += (double)v[k]; q += (double)w[k]; }
    }
    ss[g][lc] = s; qq[g][lc] = q;
    __syncthreads();
    if (g == 0 && c < C) {
      for (int k = 1; k < 8; ++k) { s += ss[k][lc]; q += qq[k][lc]; }
      result[c] = (float)s;
      result[C + c] = (float)q;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void finalize_kernel(const float* partial, float* result, int nblk, int C) {
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  __shared__ double ss[8][32], qq[8][32];
  double s = 0.0, q = 0.0;
  for (int b0 = g; b0 < nblk; b0 += 64) {
    float v[8], w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = b0 + 8 * k;
      const bool ok = b < nblk && c < C;
      v[k] = ok ? partial[((size_t)b * 2 + 0) * C + c] : 0.f;
      w[k] = ok ? partial[((size_t)b * 2 + 1) * C + c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += (double)v[k]; q += (double)w[k]; }
  }
  ss[g][threadIdx.x & 31] = s; qq[g][threadIdx.x & 31] = q;
  __syncthreads();
  if (g == 0 && c < C) {
    for (int k = 1; k < 8; ++k) { s += ss[k][threadIdx.x & 31]; q += qq[k][threadIdx.x & 31]; }
    result[c] = (float)s;
    result[C + c] = (float)q;
  }
}

int main() {
  const int C = 128, iters = 300;
  float *out, *partial, *res_a, *res_b; unsigned* ticket;
  hipMalloc(&out, (size_t)4200 * 128 * 128 * 4); hipMalloc(&partial, (size_t)4200 * 2 * C * 4);
  hipMalloc(&res_a, 2 * C * 4); hipMalloc(&res_b, 2 * C * 4); hipMalloc(&ticket, 4); hipMemset(ticket, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nblk : {138, 276, 552, 1096, 4161}) {
    float ms[6];
    for (int variant = 0; variant < 6; ++variant) {   // 0: tile kernel alone, 1: + finalize launch, 2: ticket + last workgroup, 3: fence + ticket only, 4 / 5: the same two with L2-bypassing partial rows and no fence
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) {
          tile_kernel<<<nblk, 256>>>(out, partial, ticket, res_b, variant == 2 ? 2 : variant == 3 ? 1 : variant == 4 ? 3 : variant == 5 ? 4 : 0, C);
          if (variant == 1) finalize_kernel<<<(C + 31) / 32, 256>>>(partial, res_a, nblk, C);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[variant], e0, e1);
      }
    }
    std::vector<float> a(2 * C), b(2 * C);
    hipMemcpy(a.data(), res_a, 2 * C * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), res_b, 2 * C * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2 * C; ++i) bad += a[i] != b[i];
    printf("%5d workgroups: tile kernel %.2f us | + finalize launch +%.2f | __threadfence + ticket: publish only +%.2f, with the last workgroup combining +%.2f | "
           "L2-bypassing partial rows, no fence: publish only +%.2f, combining +%.2f | results %s\n",
           nblk, 1e3 * ms[0] / iters, 1e3 * (ms[1] - ms[0]) / iters, 1e3 * (ms[3] - ms[0]) / iters, 1e3 * (ms[2] - ms[0]) / iters,
           1e3 * (ms[5] - ms[0]) / iters, 1e3 * (ms[4] - ms[0]) / iters, bad ? "DIFFER" : "equal");
  }
  return 0;
}
