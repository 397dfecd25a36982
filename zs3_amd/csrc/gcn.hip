// GCN-context cluster graph on the device: what construct_adj_mat
// (zs3/train_context_GMMN_GCNcontext.py:33-102) computes with a pure-Python depth-first search (0.3-0.5 s per
// image) -- 8-connected components of equal label numbered in raster order of their first pixel, the symmetric
// "touches in the 8-neighbourhood" adjacency between them, and each cluster's seed pixel.
//
// One workgroup per label map (the maps are at feature resolution, <= 129x129; a batch of maps is one launch, blockIdx.x =
// image, so that the training step reads all cluster counts back at once): labels live in LDS, every pass
// replaces a pixel's label by the minimum over its same-class neighbours followed by a pointer jump; the fixed
// point (label = smallest pixel index of the component) does not depend on the order in which threads run, so
// the result is deterministic.  Component roots are ranked with a block-wide prefix sum: rank = cluster id in
// raster order, exactly the numbering of the sequential search.
#include "common.h"
#include "zs3hip.h"

namespace {

constexpr int CCL_THREADS = 1024;
constexpr int CCL_MAX_PIX = 36 * 1024;   // 144 KB of LDS labels (129 x 129 = 16641 fits with room to spare)

__global__ __launch_bounds__(CCL_THREADS) void ccl_graph_kernel(const int* __restrict__ seg, int H, int W, int* __restrict__ cmap,
                                                               int* __restrict__ seed, int* __restrict__ labels,
                                                               int* __restrict__ ncluster, float* __restrict__ adj,
                                                               int cap) {
  extern __shared__ int lab[];          // [H*W] labels, then [CCL_THREADS] scan scratch
  __shared__ int changed;
  const int n = H * W, tid = threadIdx.x;
  {                                      // this block's image
    const long b = blockIdx.x;
    seg += b * n;
    cmap += b * n;
    seed += b * cap;
    labels += b * cap;
    ncluster += b;
    adj += b * cap * (long)cap;
  }
  int* scan = lab + n;
  for (int p = tid; p < n; p += CCL_THREADS) lab[p] = p;
  __syncthreads();
  for (;;) {
    if (tid == 0) changed = 0;
    __syncthreads();
    bool any = false;
    for (int p = tid; p < n; p += CCL_THREADS) {
      const int i = p / W, j = p - i * W, s = seg[p];
      int m = lab[p];
      for (int di = -1; di <= 1; ++di)
        for (int dj = -1; dj <= 1; ++dj) {
          const int a = i + di, b = j + dj;
          if (a >= 0 && a < H && b >= 0 && b < W && seg[a * W + b] == s) m = min(m, lab[a * W + b]);
        }
      m = min(m, lab[m]);   // pointer jump
      if (m < lab[p]) {
        lab[p] = m;
        any = true;
      }
    }
    if (any) changed = 1;
    __syncthreads();
    if (!changed) break;
    __syncthreads();
  }
  // rank of every root (lab[p] == p) in raster order: contiguous chunks per thread + block scan of the chunk counts
  const int per = (n + CCL_THREADS - 1) / CCL_THREADS;
  const int p0 = tid * per, p1 = min(n, p0 + per);
  int cnt = 0;
  for (int p = p0; p < p1; ++p) cnt += lab[p] == p;
  scan[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < CCL_THREADS; off <<= 1) {
    const int v = tid >= off ? scan[tid - off] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  int rank = scan[tid] - cnt;   // exclusive prefix of this chunk
  const int total = scan[CCL_THREADS - 1];
  __syncthreads();
  // overwrite root labels by -(rank+1) so that non-roots can look their cluster id up, then resolve
  for (int p = p0; p < p1; ++p)
    if (lab[p] == p) {
      if (rank < cap) {
        seed[rank] = p;
        labels[rank] = seg[p];
      }
      lab[p] = -(rank + 1);
      ++rank;
    }
  __syncthreads();
  for (int p = tid; p < n; p += CCL_THREADS) {
    const int l = lab[p];
    cmap[p] = l < 0 ? -l - 1 : -lab[l] - 1;   // a non-root's label is its root's pixel index
  }
  if (tid == 0) ncluster[0] = total;
  __syncthreads();
  __threadfence_block();
  // adjacency: clusters of different class that touch (8-neighbourhood); idempotent byte-sized facts, any order
  for (int p = tid; p < n; p += CCL_THREADS) {
    const int i = p / W, j = p - i * W, s = seg[p];
    const int c1 = cmap[p];
    if (c1 >= cap) continue;
    for (int di = -1; di <= 1; ++di)
      for (int dj = -1; dj <= 1; ++dj) {
        const int a = i + di, b = j + dj;
        if (a >= 0 && a < H && b >= 0 && b < W && seg[a * W + b] != s) {
          const int c2 = cmap[a * W + b];
          if (c2 < cap) adj[(long)c1 * cap + c2] = 1.f;
        }
      }
  }
}

}  // namespace

extern "C" int zs3_cluster_graph_max_pixels(void) { return CCL_MAX_PIX; }

extern "C" int zs3_cluster_graph_batch(const int* seg, int B, int H, int W, int* cmap, int* seed, int* labels, int* ncluster,
                                       float* adj, int cap, void* stream) {
  const int n = H * W;
  if (B < 1) return 0;
  if (n < 1 || n > CCL_MAX_PIX || cap < 1) return -1;
  const size_t lds = (size_t)(n + CCL_THREADS) * sizeof(int);
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ccl_graph_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (CCL_MAX_PIX + CCL_THREADS) * (int)sizeof(int)) != hipSuccess)
      return -4;
    configured = true;
  }
  hipLaunchKernelGGL(ccl_graph_kernel, dim3(B), dim3(CCL_THREADS), lds, (hipStream_t)stream, seg, H, W, cmap, seed, labels,
                     ncluster, adj, cap);
  return ZS3_LAUNCH_CHECK();
}

extern "C" int zs3_cluster_graph(const int* seg, int H, int W, int* cmap, int* seed, int* labels, int* ncluster, float* adj,
                                 int cap, void* stream) {
  return zs3_cluster_graph_batch(seg, 1, H, W, cmap, seed, labels, ncluster, adj, cap, stream);
}
