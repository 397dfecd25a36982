// Persistent pointwise (1x1, stride 1) convolution for gfx950 (MI355X): tile_cfg 51 (256-row tiles) / 52 (128-row tiles).
//
//   y[m, co] = epilogue( sum_ci x[m, ci] * w[co, ci] )        forward and input-gradient of the 1x1 layers
//
// The 1x1 layers of the ResNet-101 bottlenecks have short reductions (K = 256 .. 2048 channels = 8 .. 64 K steps of 32):
// in conv_igemm.hip's kernels a workgroup spends a third of its life filling its pipeline and draining its output tile,
// and every MFMA wave splits its own fp32 fragments to bf16 hi/lo (VALU-bound, MFMA pipe 28 % busy).  Here:
//
//   * A launch is min(tiles, CUs) PERSISTENT workgroups; workgroup b walks tiles b, b + G, b + 2G, ... (XCD-remapped, so
//     the workgroups of one XCD share activation rows in their L2).
//   * Four producer waves stream the K steps of ALL of the workgroup's tiles as one sequence: plain global loads three K
//     steps ahead, fp32 -> bf16 hi/lo split ONCE per element, ds_write into a two-stage LDS ring ([row][16 channels: hi 32 B |
//     lo 32 B], the swizzled 64-byte rows of conv_halo.hip; weights arrive pre-split from zs3_prep_weight).  They run
//     ahead across tile boundaries: while the MFMA waves store tile j, the first five K steps of tile j+1 are already
//     in LDS / in flight.
//   * Four consumer waves (2x2, (BM/2) x 64 each) only issue ds_read_b128 + MFMA, one workgroup barrier per K step, and
//     store their tile straight from the accumulator registers (BatchNorm partial sums, affine, activation): in the 32x32
//     MFMA layout a lane holds one column of 16 rows, so a store instruction writes two full 128-byte row segments and the
//     per-column scale / shift are per-lane scalars.  No LDS staging and nothing waited for -- the stores drain under the
//     next tile's MFMAs (the LDS-staged epilogue of conv_common.h took 17 us per 256x128 tile here, more than the tile's K
//     loop).  The barrier of the column-sum reduction is matched by the producers (s_barrier counts waves, not meanings),
//     which keep their loads in flight across it.
//   * Epilogues that have to LOAD per element (residual, accumulate, the fused BN-backward sums of the dgrad launches) would
//     stall the only MFMA waves of the CU on memory latency: those launches stay on conv_igemm.hip's kernels, whose
//     several workgroups per CU hide it (zs3_conv_igemm returns -7 for them on tile_cfg 51 / 52).
// Replaces F.conv2d of the 1x1 nn.Conv2d at resnet.py:33-53 (conv1, conv3), aspp.py:86-88 and their input gradients.
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"
#include "zs3hip.h"

namespace {

constexpr int PW_BN = 128;
constexpr int PW_BSTAGE = 2 * 8192;                 // weights of a K step: two 16-channel sub-chunks x 128 columns x 64 B
constexpr int PW_CTILE = 4 * PW_BN * 4;             // the epilogue's cross-wave column sums (BatchNorm partials)
// -DZS3_PW_ABLATE=n builds (tools/probe/build_variant.sh; timing probes, wrong results): 1 = no epilogue work (barriers kept),
// 2 = no MFMAs, 4 = no global loads, 8 = no split / LDS writes, 32 = no BatchNorm sums
#ifndef ZS3_PW_ABLATE
#define ZS3_PW_ABLATE 0
#endif
int g_pw_wgs = 256;                                 // persistent workgroups per launch (zs3_conv_pw_set_wgs)

// INAFF: the producers apply x' = max(x * in_scale[c] + in_shift[c], 0) before the split (conv_common.h: ConvArgs::in_scale).
template <int PREC, int BM, bool INAFF = false>
__global__ __launch_bounds__(512) void conv_pw_kernel(const ConvArgs p, const int ntiles, const int ntn) {
  constexpr int BN = PW_BN, TM = BM / 64, TN = 2;
  constexpr int ASUB = BM * 64;                     // one 16-channel sub-chunk of the activation rows
  constexpr int STAGE = 2 * ASUB + PW_BSTAGE;
  constexpr int OFF_CT = 2 * STAGE;
  constexpr int NL = PREC == 3 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int first = xcd_remap(blockIdx.x, G);       // G <= ntiles: every workgroup owns at least one tile
  const int nmine = (ntiles - first + G - 1) / G;
  const int NK = p.cin_pad >> 5;
  const int S = nmine * NK;                         // K steps of this workgroup
  const int S3 = (S + 2) / 3 * 3;                   // the producers' schedule is unrolled by three (register sets)
  const int E = p.stat_partial ? 1 : 0;   // workgroup barriers of one epilogue

  if (wave >= 4) {
    // ------------------------------------------------------------------ producers (256 lanes)
    // activations: lane -> (row prow + 32 r, channels 4 c8 .. +3 of the K step): 8 lanes read one row's 128 contiguous bytes
    // weights    : lane -> (columns wrow, wrow + 64; 16-byte piece wq of both sub-chunks)
    constexpr int NRA = BM / 32;
    const int pl = tid - 256, prow = pl >> 3, c8 = pl & 7, sub = c8 >> 2, cq = c8 & 3;
    const int wrow = pl >> 2, wq = pl & 3;
    const int qoff = (wq & 1) * 16 + (wq >> 1) * 64;
    const unsigned wdst = (unsigned)(wrow * 64 + ((wq ^ ((wrow >> 2) & 3)) << 4));
    const long rstep = 32 * (long)p.ldx;
    // load stream state: tile / K step of the next request
    int lt = first, lk = 0;
    const float* abase = nullptr;
    unsigned rowmask = 0u;
    const unsigned char* wptr[2];
    int wstep[2];
    auto setup_tile = [&](int tile) {
      tile = tile < ntiles ? tile : first + (nmine - 1) * G;   // past the end: re-request the last tile (never multiplied)
      const int mt = tile / ntn, nt = tile - mt * ntn;
      const int m0 = mt * BM, n0 = nt * BN;
      abase = p.x + (long)(m0 + prow) * p.ldx + c8 * 4;
      rowmask = 0u;
#pragma unroll
      for (int r = 0; r < NRA; ++r) rowmask |= (m0 + prow + 32 * r < p.M ? 1u : 0u) << r;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int col = n0 + wrow + 64 * e;
        const bool ok = col < p.ncols;
        wptr[e] = ok ? reinterpret_cast<const unsigned char*>(p.w_pk) + (size_t)col * (4 * (size_t)p.ldw) + qoff
                     : reinterpret_cast<const unsigned char*>(p.zero);
        wstep[e] = ok ? 1 : 0;
      }
    };
    struct StepRegs {
      f32x4 a[NRA];
      u32x4 w[2][2];
      f32x4 sc, sh;      // (INAFF) scale / shift of this lane's four channels in this K step
    };
    StepRegs buf[3];
    auto load_step = [&](StepRegs& d) {
      const bool cok = lk * 32 + c8 * 4 < p.cin_valid;
      const float* src = abase + lk * 32;
      if (!(ZS3_PW_ABLATE & 4)) {
#pragma unroll
        for (int r = 0; r < NRA; ++r) {
          const float* s = (cok && ((rowmask >> r) & 1u)) ? src + r * rstep : p.zero;
          d.a[r] = *reinterpret_cast<const f32x4*>(s);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            d.w[e][s] = *reinterpret_cast<const u32x4*>(wptr[e] + (size_t)(lk * 128 + s * 32) * wstep[e]);
        if constexpr (INAFF) {
          d.sc = *reinterpret_cast<const f32x4*>(cok ? p.in_scale + lk * 32 + c8 * 4 : p.zero);
          d.sh = *reinterpret_cast<const f32x4*>(cok ? p.in_shift + lk * 32 + c8 * 4 : p.zero);
        }
      }
      if (++lk == NK) {
        lk = 0;
        lt += G;
        setup_tile(lt);
      }
    };
    auto write_step = [&](const StepRegs& s, int stage) {
      if (ZS3_PW_ABLATE & 8) return;
      unsigned char* sa = dsm + stage * STAGE + sub * ASUB;
#pragma unroll
      for (int r = 0; r < NRA; ++r) {
        const int row = prow + 32 * r, sw = (row >> 2) & 3;
        u32x2 hi, lo;
        unsigned h, l;
        f32x4 v = s.a[r];
        if constexpr (INAFF) {
          // (rows past M are loaded as zeros and become relu(shift) here: their outputs are never stored, and the BatchNorm sums of
          // the one partial tile skip them in the epilogue -- cheaper than four selects per row in every tile)
          v = affine_relu4(v, s.sc, s.sh);
        }
        split_pair<PREC>(v[0], v[1], h, l); hi[0] = h; lo[0] = l;
        split_pair<PREC>(v[2], v[3], h, l); hi[1] = h; lo[1] = l;
        const int o = (((cq >> 1) ^ sw) << 4) + (cq & 1) * 8;
        *reinterpret_cast<u32x2*>(sa + row * 64 + o) = hi;
        if (PREC == 3) *reinterpret_cast<u32x2*>(sa + row * 64 + (o ^ 32)) = lo;
      }
      unsigned char* sb = dsm + stage * STAGE + 2 * ASUB;
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c) *reinterpret_cast<u32x4*>(sb + c * 8192 + e * 4096 + wdst) = s.w[e][c];
    };
    // prologue: step 0 in stage 0; steps 1, 2, 3 requested into register sets 1, 2, 0
    setup_tile(first);
    load_step(buf[0]);
    write_step(buf[0], 0);
    load_step(buf[1]);
    load_step(buf[2]);
    load_step(buf[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // B_0
    // interval i (the consumers multiply step i): write step i + 1 into the other stage, request step i + 4.  An interval
    // that starts a tile (i = NK, 2 NK, ...) also takes part in the E barriers of the previous tile's epilogue.
    int nexttile = NK;
    for (int i0 = 0; i0 < S3; i0 += 3) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int i = i0 + r;
        write_step(buf[(r + 1) % 3], (i + 1) & 1);
        load_step(buf[(r + 1) % 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i == nexttile) {
          nexttile += NK;
          if (i < S)
            for (int e = 0; e < E; ++e) __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();   // B_{i+1}: step i + 1 is in LDS
      }
    }
    for (int e = 0; e < E; ++e) __builtin_amdgcn_s_barrier();   // the last tile's epilogue
  } else {
    // ------------------------------------------------------------------ consumers: ds_read_b128 + MFMA, then the epilogue
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int lr = lane & 31, kh = lane >> 5;
    const unsigned swz = (unsigned)((kh ^ ((lr >> 2) & 3)) << 4);
    const unsigned aoff = (unsigned)((wm * (BM / 2) + lr) * 64) + swz;            // row block i: + 2048 i; sub-chunk: + ASUB
    const unsigned boff = (unsigned)(2 * ASUB + (wn * 64 + lr) * 64) + swz;       // column block j: + 2048 j; sub-chunk: + 8192
    f32x16 acc[TM][TN];
    bf16x8 fa[2][TM][2], fb[2][TN][2];   // [register set][block][hi, lo]
    constexpr int NREAD = (TM + TN) * NL;
    constexpr int NMF = TM * TN * (PREC == 3 ? 3 : 1);
    constexpr int RS = NMF / NREAD >= 1 ? NMF / NREAD : 1;      // one fragment read every RS MFMAs
    constexpr int RPER = (NREAD + NMF - 1) / NMF;               // (plain bf16: more reads than MFMAs)
    auto read_k = [&](auto setc, auto kc, const unsigned char* st, int sc) {
      constexpr int SET = decltype(setc)::value, K = decltype(kc)::value;
      if constexpr (K < NREAD) {
        constexpr int blk = K / NL, pl = K % NL;
        if constexpr (blk < TM)
          fa[SET][blk][pl] = *reinterpret_cast<const bf16x8*>(st + ((aoff + blk * 2048 + sc * ASUB) ^ (32u * pl)));
        else
          fb[SET][blk - TM][pl] = *reinterpret_cast<const bf16x8*>(st + ((boff + (blk - TM) * 2048 + sc * 8192) ^ (32u * pl)));
      }
    };
    // one 16-channel sub-step from register set SET; the other set is filled from sub-chunk `sc` of stage `st` meanwhile
    auto substep = [&](auto setc, const unsigned char* st, int sc) {
      constexpr int SET = decltype(setc)::value;
      auto mf = [&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int pr = m / (TM * TN), i = (m / TN) % TM, j = m % TN;
        constexpr int ia = PREC == 3 ? (pr == 0 ? 1 : 0) : 0, ib = PREC == 3 ? (pr == 1 ? 1 : 0) : 0;
        if (!(ZS3_PW_ABLATE & 2))
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i][ia], fb[SET][j][ib], acc[i][j], 0, 0, 0);
        if constexpr (RPER == 1) {
          if constexpr (m % RS == 0) read_k(std::integral_constant<int, SET ^ 1>{}, std::integral_constant<int, m / RS>{}, st, sc);
        } else {
          read_k(std::integral_constant<int, SET ^ 1>{}, std::integral_constant<int, m * RPER>{}, st, sc);
          read_k(std::integral_constant<int, SET ^ 1>{}, std::integral_constant<int, m * RPER + 1>{}, st, sc);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto run = [&](auto self, auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m < NMF) {
          mf(mc);
          self(self, std::integral_constant<int, m + 1>{});
        }
      };
      run(run, std::integral_constant<int, 0>{});
    };
    __builtin_amdgcn_s_barrier();   // B_0
    asm volatile("" ::: "memory");
    {
      auto fill = [&](auto self, auto kc) {
        constexpr int K = decltype(kc)::value;
        if constexpr (K < NREAD) {
          read_k(std::integral_constant<int, 0>{}, kc, dsm, 0);
          self(self, std::integral_constant<int, K + 1>{});
        }
      };
      fill(fill, std::integral_constant<int, 0>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    float* const ctile = reinterpret_cast<float*>(dsm + OFF_CT);
    auto lds_barrier = [&]() {   // the epilogue's hazards are on the staging area only: do not drain the global stores
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    int gs = 0;   // global K-step counter (stage = gs & 1)
    for (int tj = 0; tj < nmine; ++tj) {
      const int tile = first + tj * G;
      const int mt = tile / ntn, nt = tile - mt * ntn;
      const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      for (int k = 0; k < NK; ++k, ++gs) {
        const unsigned char* cur = dsm + (gs & 1) * STAGE;
        const unsigned char* nxt = dsm + ((gs + 1) & 1) * STAGE;
        substep(std::integral_constant<int, 0>{}, cur, 1);   // channels 0..15 of the step; fetch 16..31
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // B_{gs+1}: the other stage holds step gs + 1
        asm volatile("" ::: "memory");
        substep(std::integral_constant<int, 1>{}, nxt, 0);   // channels 16..31; fetch the next step's 0..15
      }
      // ---- epilogue of this tile (consumer waves only; the producers are already filling the next tile's stages)
      if (ZS3_PW_ABLATE & 1) {
        for (int e = 0; e < E; ++e) __builtin_amdgcn_s_barrier();
        if (m0 < 0) p.y[tid] = acc[0][0][0] + acc[TM - 1][1][5];
        continue;
      }
      // straight from the accumulator registers; only the per-column BatchNorm sums cross waves (2 KB of LDS, one barrier)
      if ((ZS3_PW_ABLATE & 32) && p.stat_partial) {
        lds_barrier();
      } else if (p.stat_partial) {
        float* red = ctile;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float s = 0.f, q2 = 0.f;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = acc[i][j][r];
              if (INAFF && m0 + BM > p.M && m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) >= p.M) v = 0.f;
              s += v;
              q2 = fmaf(v, v, q2);
            }
          s += __shfl_xor(s, 32, 64);
          q2 += __shfl_xor(q2, 32, 64);
          if (lane < 32) {
            red[(wm * 2 + 0) * BN + wn * 64 + j * 32 + lane] = s;
            red[(wm * 2 + 1) * BN + wn * 64 + j * 32 + lane] = q2;
          }
        }
        lds_barrier();
        if (tid < BN) {
          const int col = n0 + tid;
          if (col < p.ncols) {
            p.stat_partial[((size_t)mt * 2 + 0) * p.ncols + col] = red[tid] + red[2 * BN + tid];
            p.stat_partial[((size_t)mt * 2 + 1) * p.ncols + col] = red[BN + tid] + red[3 * BN + tid];
          }
        }
      }
      store_acc_direct<TM, TN, BM, BN>(p, acc, m0, n0, wm, wn, lane);
    }
    for (int e = S; e < S3; ++e) __builtin_amdgcn_s_barrier();   // the producers' schedule is padded to a multiple of three
  }
}

bool pw_ok(const ConvArgs& a, int bm) {
  if (!direct_epilogue(a)) return false;   // epilogues that load per element
  if (a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad_h != 0 || a.pad_w != 0 || a.H != a.Ho || a.W != a.Wo) return false;
  if ((a.ldx & 3) || (a.cin_valid & 3) || (a.cin_pad & 31) || a.cin_pad < 32 || a.M <= 0) return false;
  return bm == 256 || bm == 128;
}

template <int PREC, int BM, bool INAFF = false>
int launch_pw_t(const ConvArgs& a, hipStream_t st) {
  static bool configured = false;
  constexpr int LDS = 2 * (2 * BM * 64 + PW_BSTAGE) + PW_CTILE;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_kernel<PREC, BM, INAFF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return -4;
    configured = true;
  }
  const int ntn = (a.ncols + PW_BN - 1) / PW_BN;
  const int ntiles = ((a.M + BM - 1) / BM) * ntn;
  const int grid = ntiles < g_pw_wgs ? ntiles : g_pw_wgs;
  hipLaunchKernelGGL((conv_pw_kernel<PREC, BM, INAFF>), dim3(grid), dim3(512), LDS, st, a, ntiles, ntn);
  return ZS3_LAUNCH_CHECK();
}

}  // namespace

int zs3conv::pw_eligible(const ConvArgs& a, int bm) { return pw_ok(a, bm) ? 1 : 0; }

int zs3conv::launch_pw(const ConvArgs& a, int bm, int prec, hipStream_t st) {
  if (!pw_ok(a, bm)) return -7;
  if (a.in_scale) {   // input transform (the producing layer's BatchNorm-apply + ReLU) in the producer waves
    if (!a.in_shift) return -1;
    if (bm == 256) return prec == 1 ? launch_pw_t<1, 256, true>(a, st) : launch_pw_t<3, 256, true>(a, st);
    return prec == 1 ? launch_pw_t<1, 128, true>(a, st) : launch_pw_t<3, 128, true>(a, st);
  }
  if (bm == 256) return prec == 1 ? launch_pw_t<1, 256>(a, st) : launch_pw_t<3, 256>(a, st);
  return prec == 1 ? launch_pw_t<1, 128>(a, st) : launch_pw_t<3, 128>(a, st);
}

// Whether tile_cfg 51 / 52 can run this convolution (1x1, stride 1, no padding; callers fall back to tile_cfg 31 otherwise).
extern "C" int zs3_conv_pw_ok(int N, int H, int W, int Ho, int Wo, int cin_pad, int cin_valid, int ldx, int KH, int KW, int stride,
                              int pad_h, int pad_w, int tile_cfg) {
  ConvArgs a{};
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.cin_pad = cin_pad; a.cin_valid = cin_valid; a.ldx = ldx;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  a.M = N * Ho * Wo;
  return zs3conv::pw_eligible(a, tile_cfg == 52 ? 128 : 256);
}

// Persistent workgroups per launch of tile_cfg 51 / 52 (default 256 = one per CU); returns the previous value.  Tests set a
// small number so that small problems exercise the cross-tile pipeline.
extern "C" int zs3_conv_pw_set_wgs(int wgs) {
  const int old = g_pw_wgs;
  if (wgs > 0) g_pw_wgs = wgs;
  return old;
}
