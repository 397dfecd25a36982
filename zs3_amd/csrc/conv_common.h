// Shared pieces of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_halo.hip): the launch arguments and
// the fused output-tile epilogue (affine / residual / activation / accumulate, BatchNorm partial sums).
#pragma once
#include <type_traits>

#include "common.h"

namespace zs3conv {

struct ConvArgs {
  const float* x;
  const unsigned short* w_pk;   // [ncols][K/32][{hi,lo}][32] bf16 (zs3_prep_weight)
  float* y;
  const float* scale;
  const float* shift;
  const float* res;
  float* stat_partial;
  const float* zero;   // >= 256 B of zeros: source of every masked (padding / out-of-range) load
  int N, H, W, Ho, Wo;
  int cin_pad, cin_valid, ldx;
  int KH, KW, stride, pad_h, pad_w, dil;
  int ncols, ldw, ldy, ldr, M;
  int act, accumulate, dgrad, stride_log2;
  // optional transform of the input on its way into LDS (the producing layer's BatchNorm-apply + ReLU, so that its activation
  // tensor is never stored): x' = max(x * in_scale[c] + in_shift[c], 0).  Strip-resident and persistent pointwise kernels only.
  const float* in_scale;
  const float* in_shift;
  int x_bf16;   // x is stored as bf16 (`io` bit 0; ldx counts elements)
  int y_bf16;   // y -- and with it res, the accumulate target and bs_y: the layer-level activations of the output side -- are bf16 (`io` bit 1)
  float leak;
  // optional BatchNorm-backward statistics of the layer whose output gradient this launch produces (dgrad epilogue):
  // bs_partial[mtile][2][ncols] = (sum dz, sum dz*xhat) with dz = stored value * ReLU mask, xhat = (bs_y - mean) * istd
  const float* bs_y;
  const float* bs_mean;
  const float* bs_istd;
  const float* bs_msc;            // mask = bs_y * msc + msh > 0 (no residual) ...
  const float* bs_msh;
  const unsigned char* bs_mbits;  // ... or sign bits [M][ncols/4] (residual layers)
  float* bs_partial;
  int bs_ldy;
  // optional ReLU mask applied to `res` before it is added (sign bytes [M][ncols/4]): the skip gradient of a residual
  // block is (block-output gradient) * mask, taken straight from the block-output gradient instead of a stored copy
  const unsigned char* res_mbits;
};

// conv_halo.hip: strip-resident 3x3 (any multi-tap, stride-1, same-size) convolution, tile_cfg 41 (256-row tiles) / 42 (192).
// `bm` = 0 asks whether the launch is eligible at all (returns 1 / 0); otherwise launches and returns the HIP status.
int halo_eligible(const ConvArgs& a, int bm, int prec);
int launch_halo(const ConvArgs& a, int bm, int prec, hipStream_t st);
// conv_pw.hip: persistent pointwise (1x1, stride 1) convolution, tile_cfg 51 (256-row tiles) / 52 (128-row tiles).
int pw_eligible(const ConvArgs& a, int bm);
int launch_pw(const ConvArgs& a, int bm, int prec, hipStream_t st);

}  // namespace zs3conv
using zs3conv::ConvArgs;

namespace {

// Rows [0, nrows) of an LDS-staged output tile -> global memory as dwordx4 per lane, with the fused epilogue (affine,
// residual, activation, accumulate).  `c4`/`r0` = this thread's column quad / first row, RPP = rows per pass.  When
// p.bs_partial is set the thread also accumulates the BN-backward sums of its 4 columns over the rows it stores.
//
// Round 4: epilogues that LOAD per element (residual, accumulate, the BN-backward operand: every data-gradient launch) fetch the
// operands of EPI_U rows at once through branch-free addresses (an absent operand or a row past M reads the zero page) and
// only then combine and store.  The round-3 form loaded each operand under `if (p.res) ...` inside the row loop, and hipcc waits
// vmcnt(0) wherever a branch that holds a load joins: its ISA was `load, wait, store` per row -- up to 16 dependent HBM round
// trips per thread and tile (the strip-resident kernel's data-gradient launches: tools/probe, the L / vmcnt pattern of conv_halo.s).
// Y16: the element type of y / res / bn_y is a template argument for the same reason (two load widths under one run-time test).
template <int RPP, bool Y16, int EPI_U>
__device__ __forceinline__ void store_tile_rows_t(const ConvArgs& p, const float* ctile, int ldc, int row_base, int nrows,
                                                  int col, int c4, int r0, f32x4 sc, f32x4 sh, bool affine, bool vec,
                                                  f32x4& bs_s, f32x4& bs_q) {
  using YT = std::conditional_t<Y16, bf16_t, float>;
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {0.f, 0.f, 0.f, 0.f}, msc = {0.f, 0.f, 0.f, 0.f}, msh = {0.f, 0.f, 0.f, 0.f};
  const bool bstat = p.bs_partial != nullptr && vec && col < p.ncols;
  if (bstat) {
    mu = *reinterpret_cast<const f32x4*>(p.bs_mean + col);
    is = *reinterpret_cast<const f32x4*>(p.bs_istd + col);
    if (p.bs_msc) {
      msc = *reinterpret_cast<const f32x4*>(p.bs_msc + col);
      msh = *reinterpret_cast<const f32x4*>(p.bs_msh + col);
    }
  }
  YT* const yp = reinterpret_cast<YT*>(p.y);
  auto act_of = [&](f32x4 v) {
    if (p.act == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.act == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.leak;
    }
    return v;
  };
  const bool loading = vec && (p.res != nullptr || p.accumulate || bstat);
  if (vec && !loading) {
    // store-only epilogue (every forward layer): nothing to wait for
    for (int rr = r0; rr < nrows; rr += RPP) {
      const int row = row_base + rr;
      if (row >= p.M || col >= p.ncols) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(ctile + rr * ldc + c4 * 4);
      if (affine) v = v * sc + sh;
      st4<YT>(yp + (size_t)row * p.ldy + col, act_of(v));
    }
    return;
  }
  if (loading) {
    const YT* const zero = reinterpret_cast<const YT*>(p.zero);
    const unsigned char* const zb = reinterpret_cast<const unsigned char*>(p.zero);
    const YT* const resp = reinterpret_cast<const YT*>(p.res);
    const YT* const byp = reinterpret_cast<const YT*>(p.bs_y);
    const int mstride = p.ncols >> 2, mcol = col >> 2;
    for (int rr0 = r0; rr0 < nrows; rr0 += EPI_U * RPP) {
      f32x4 rv[EPI_U], ov[EPI_U], yv[EPI_U];
      unsigned rmb[EPI_U], bmb[EPI_U];
      bool ok[EPI_U];
      // ---- every operand of EPI_U rows requested before any of them is used; selects, not branches
#pragma unroll
      for (int u = 0; u < EPI_U; ++u) {
        const int rr = rr0 + u * RPP, row = row_base + rr;
        ok[u] = rr < nrows && row < p.M && col < p.ncols;
        const size_t rowc = ok[u] ? (size_t)row : 0;
        rv[u] = ld4<YT>((ok[u] && resp) ? resp + rowc * p.ldr + col : zero);
        ov[u] = ld4<YT>((ok[u] && p.accumulate) ? yp + rowc * p.ldy + col : zero);
        yv[u] = ld4<YT>((ok[u] && bstat) ? byp + rowc * p.bs_ldy + col : zero);
        rmb[u] = *((ok[u] && p.res_mbits) ? p.res_mbits + rowc * mstride + mcol : zb);
        bmb[u] = *((ok[u] && p.bs_mbits) ? p.bs_mbits + rowc * mstride + mcol : zb);
      }
#pragma unroll
      for (int u = 0; u < EPI_U; ++u) {
        const int rr = rr0 + u * RPP, row = row_base + rr;
        if (!ok[u]) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(ctile + rr * ldc + c4 * 4);
        if (affine) v = v * sc + sh;
        if (p.res) {
          f32x4 r = rv[u];
          if (p.res_mbits) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (rmb[u] >> e) & 1u ? r[e] : 0.f;
          }
          v = v + r;
        }
        v = act_of(v);
        if (p.accumulate) v = v + ov[u];
        if constexpr (Y16) {
          // the BatchNorm-backward sums below are taken over the values the next kernel will read: the stored, rounded ones
          const u32x2 pk = f32x4_to_bf16(v);
          *reinterpret_cast<u32x2*>(yp + (size_t)row * p.ldy + col) = pk;
          v = bf16x4_to_f32(pk);
        } else {
          *reinterpret_cast<f32x4*>(yp + (size_t)row * p.ldy + col) = v;
        }
        if (bstat) {
          f32x4 dz = v;
          if (p.bs_mbits) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[e] = (bmb[u] >> e) & 1u ? dz[e] : 0.f;
          } else if (p.bs_msc) {
            const f32x4 av = yv[u] * msc + msh;
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[e] = av[e] > 0.f ? dz[e] : 0.f;
          }
          bs_s += dz;
          bs_q += dz * ((yv[u] - mu) * is);
        }
      }
    }
    return;
  }
  // scalar tail form: channel counts / strides that are not multiples of four (the 21- and 60-class heads)
  for (int rr = r0; rr < nrows; rr += RPP) {
    const int row = row_base + rr;
    if (row >= p.M || col >= p.ncols) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(ctile + rr * ldc + c4 * 4);
    if (affine) v = v * sc + sh;
    const size_t di = (size_t)row * p.ldy + col;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col + e < p.ncols) {
        float t = v[e];
        if (p.res) t += ld1<YT>(reinterpret_cast<const YT*>(p.res) + (size_t)row * p.ldr + col + e);
        if (p.act == 1) t = fmaxf(t, 0.f);
        else if (p.act == 2) t = t > 0.f ? t : t * p.leak;
        if (p.accumulate) t += ld1<YT>(yp + di + e);
        st1<YT>(yp + di + e, t);
      }
  }
}
// EPI_U: rows whose operands are in flight together (4; 2 where the caller still holds live accumulators: conv_halo.hip's 256-row tiles)
template <int RPP, int EPI_U = 4>
__device__ __forceinline__ void store_tile_rows(const ConvArgs& p, const float* ctile, int ldc, int row_base, int nrows,
                                                int col, int c4, int r0, f32x4 sc, f32x4 sh, bool affine, bool vec,
                                                f32x4& bs_s, f32x4& bs_q) {
  if (p.y_bf16) store_tile_rows_t<RPP, true, EPI_U>(p, ctile, ldc, row_base, nrows, col, c4, r0, sc, sh, affine, vec, bs_s, bs_q);
  else store_tile_rows_t<RPP, false, EPI_U>(p, ctile, ldc, row_base, nrows, col, c4, r0, sc, sh, affine, vec, bs_s, bs_q);
}

// Block reduction of the per-thread BN-backward sums (fixed order: deterministic) and store of this row tile's partials.
// Thread (r0, c4) holds the sums of columns 4*c4..4*c4+3 over its rows; `red` needs 2 * RPP * BN floats of LDS.
template <int BN, int RPP, int NT>
__device__ __forceinline__ void finish_bwd_stats(const ConvArgs& p, float* red, int tid, int c4, int r0, int mt, int n0,
                                                 f32x4 bs_s, f32x4 bs_q, float* dst = nullptr) {
  if (!dst) dst = p.bs_partial;
  __syncthreads();   // the output tile in LDS is no longer read
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[r0 * BN + c4 * 4 + e] = bs_s[e];
    red[(RPP + r0) * BN + c4 * 4 + e] = bs_q[e];
  }
  __syncthreads();
  for (int c = tid; c < BN; c += NT) {
    const int col = n0 + c;
    if (col < p.ncols) {
      float ts = 0.f, tq = 0.f;
      for (int g = 0; g < RPP; ++g) {
        ts += red[g * BN + c];
        tq += red[(RPP + g) * BN + c];
      }
      dst[((size_t)mt * 2 + 0) * p.ncols + col] = ts;
      dst[((size_t)mt * 2 + 1) * p.ncols + col] = tq;
    }
  }
}

// Epilogues that only STORE (raw output, or affine + activation: the forward layers) can leave the accumulator registers
// directly: in the 32x32 MFMA layout a lane holds one column (lane & 31) of 16 rows, so one store instruction writes two full
// 128-byte row segments and the per-column scale / shift are per-lane scalars -- no LDS staging, no barrier, nothing waited
// for.  Used by the persistent kernel (conv_pw.hip), whose four MFMA waves are alone on their CU: the staged path above cost them
// ~14 us per 256x128 tile, this one ~5.  (conv_halo.hip's eight-wave staged epilogue gains nothing from it: 4.77 vs 4.76 ms over
// the 3x3 forward layers, 46.5 vs 46.4 ms per step -- tools/probe/r3o.sh.)
__host__ __device__ __forceinline__ bool direct_epilogue(const ConvArgs& p) {
  return !p.res && !p.accumulate && !p.bs_partial && !p.res_mbits;
}

// Wave (wm, wn) of a 2x2 consumer grid holds rows wm*BM/2 + 32 i + .., columns wn*64 + 32 j + .. of the BM x BN tile at (m0, n0).
template <int TM, int TN, int BM, int BN>
__device__ __forceinline__ void store_acc_direct(const ConvArgs& p, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                                 int lane) {
  const int hh = lane >> 5, lr = lane & 31;
  const bool full = m0 + BM <= p.M;              // no row tail in this tile (wave-uniform)
  const bool affine = (p.scale != nullptr) || (p.shift != nullptr);
  // element (i, j, r) of this lane: row m0 + lrow + 32 i + rofs(r), column n0 + lcol + 32 j; a uniform tile base plus a 32-bit
  // lane offset
  int opaque = 0;
  asm volatile("" : "+v"(opaque));   // hipcc otherwise computes the 128 store addresses before the K loop and spills them
  const int lrow = wm * (BM / 2) + 4 * hh + opaque, lcol = wn * 64 + lr;
  float* const ybase = p.y + (size_t)m0 * p.ldy + n0;
  // whole tiles (no row / column tail) in the three epilogue forms the network uses take the lean path: per element one v_add
  // (uniform row offset + lane offset), optionally fma / max, one store -- mode tests per element cost 30 instructions
  auto store_fast = [&](auto affc, auto reluc) {
    constexpr bool AFF = decltype(affc)::value, RELU = decltype(reluc)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + lcol + 32 * j;
      const float sc = AFF && p.scale ? p.scale[col] : 1.f, sh = AFF && p.shift ? p.shift[col] : 0.f;
      const unsigned lb = (unsigned)(lrow * p.ldy + lcol + 32 * j) * 4u;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned ub = (unsigned)((32 * i + (r & 3) + 8 * (r >> 2)) * p.ldy) * 4u;
          float v = acc[i][j][r];
          if (AFF) v = fmaf(v, sc, sh);
          if (RELU) v = fmaxf(v, 0.f);
          *reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ybase) + (lb + ub)) = v;
        }
    }
  };
  const bool whole = full && n0 + BN <= p.ncols && (size_t)BM * p.ldy < (1u << 28);
  if (whole && !affine && p.act == 0) {
    store_fast(std::false_type{}, std::false_type{});
  } else if (whole && affine && p.act == 1) {
    store_fast(std::true_type{}, std::true_type{});
  } else if (whole && affine && p.act == 0) {
    store_fast(std::true_type{}, std::false_type{});
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + lcol + 32 * j;
      const bool cok = col < p.ncols;
      const int cc = cok ? col : 0;
      const float sc = p.scale ? p.scale[cc] : 1.f, sh = p.shift ? p.shift[cc] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int ro = lrow + 32 * i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = ro + (r & 3) + 8 * (r >> 2);
          float v = acc[i][j][r];
          if (affine) v = fmaf(v, sc, sh);
          if (p.act == 1) v = fmaxf(v, 0.f);
          else if (p.act == 2) v = v > 0.f ? v : v * p.leak;
          if (cok && (full || m0 + rr < p.M)) ybase[rr * p.ldy + lcol + 32 * j] = v;
        }
      }
    }
  }
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

}  // namespace
