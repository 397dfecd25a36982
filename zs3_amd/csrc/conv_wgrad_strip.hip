// Strip-resident weight gradient of the stride-1, same-size 3x3 (dilated) convolutions for gfx950 (MI355X).
//
//   dw[co, t, ci] = sum_p dy[p, co] * x[p + tap_t, ci]                          (zs3_conv_wgrad_strip)
//
// conv_wgrad.hip computes this as one 256x256 GEMM per filter tap over the pixel axis: both operands are fp32
// activations, every wave gathers and splits its own fragments (6 VALU per MFMA -- the kernel is VALU-bound at ~157 TF),
// and x is re-read once per tap.  Here:
//
//   * The reduction runs over ZERO-PADDED pixel positions q = (n*(H+d) + y)*(W+d) + x: every image row carries d pad
//     columns and every image d pad rows (shared with the next row / image), where dy and x are zero.  In that space tap
//     (a, b) is the constant offset a*d*(W+d) + b*d for EVERY position -- border taps read pad zeros by themselves, so
//     there are no masks, no per-pixel coordinates and no divergence in the multiply loop.  Cost: (H+d)(W+d)/(HW) more
//     K steps (6 % at 33x33, 1.6 % at 129x129).
//   * A workgroup owns a 64 (co) x 64 (ci) tile of ALL NINE taps and a range of K steps (split-K).  Per 16-position K
//     step it needs 16 new rows of dy and 16 new rows of x: x lives in an LDS ring that holds the positions every tap's
//     shifted window can reach (2d(W+d+1) + 16 rows), so x is read from L2 once instead of nine times.
//   * Four producer waves load both operands as fp32 (plain loads, three K steps ahead), split them to bf16 hi/lo ONCE
//     and write [position][channel] rows (hi 128 B | lo 128 B | 64 B pad = 320 B: four consecutive rows cover the 64
//     banks) -- the conversion has left the MFMA waves.
//   * Four consumer waves (2x2 blocks of 32 co x 32 ci, nine taps each = 144 accumulator registers) fetch their MFMA
//     operands with ds_read_b64_tr_b16: the operand wants 8 consecutive POSITIONS per lane while memory is channel-
//     contiguous, and the transposing read delivers exactly that from the row-major image (16 lanes read a 4-position x
//     16-channel block, lane = channel gets the 4 positions; measured semantics: result[l][j] = source[4j + l/4][l%4]
//     within a 16-lane group, tools/probe/tr/tr_host.hip) -- no packing, no VALU.  27 MFMAs per wave and K step against
//     40 LDS reads and a few scalar window computations.
//
// The first 16 ring rows are mirrored behind the ring so that a 16-row window never wraps.  Partial sums of the split-K
// ranges go to slabs that zs3_conv_wgrad_strip reduces in a fixed order (deterministic).
// Replaces convolution_backward(weight) of the 3x3 nn.Conv2d of resnet.py:18-26, aspp.py:11-19, decoder.py:15-24.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "zs3hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct StripArgs {
  const float* dy;
  const float* x;
  float* out;          // slab 0 (or dw itself when splitk == 1)
  const float* zero;
  int N, H, W;
  int co_read, co_write, ci_read, ci_write;
  int lddy, ldx, ldw, cin_w;
  int Hd, Wd;          // padded extents H + d, W + d
  long Q;              // padded positions N * Hd * Wd
  int RL;              // positions the x stream starts before the first dy position (multiple of 16, >= the tap reach R)
  int lead;            // K step i reads x steps i .. i + lead; (lead + 1) % 3 == 0
  int ring_steps;      // lead + 2
  int steps_per_split; // multiple of 6
  int tiles_co, tiles_ci;
  long slab;           // elements per split-K slab
  int toff[9];         // tap offsets in padded positions
  // optional transform of x on its way into LDS: x' = max(x * x_scale[c] + x_shift[c], 0) at real positions (pad positions stay 0)
  // -- the BatchNorm-apply + ReLU of the layer that produced x, whose activation tensor is then never stored
  const float* x_scale;
  const float* x_shift;
};

constexpr int WS_ROW = 320;            // bytes per LDS row: 64 ch hi | 64 ch lo | 64 B pad

struct Pos {           // a lane's position in a padded stream: q and its coordinates (valid while 0 <= q)
  long q;
  int n, yp, xp;
};

// IO16: dy and x are stored as bf16 (plain-bf16 products only): the producers copy 8 bytes per lane and position into the LDS rows,
// no conversion (a register set then holds the raw pair of words in its first two components).
template <int PREC, bool XAFF = false, bool IO16 = false>
__global__ __launch_bounds__(512) void conv_wgrad_strip_kernel(const StripArgs p) {
  static_assert(!IO16 || (PREC == 1 && !XAFF), "bf16-stored operands: plain bf16 products, no producer-side transform");
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int ntiles = p.tiles_co * p.tiles_ci;
  const int split = b / ntiles;
  b -= split * ntiles;
  const int tco = b / p.tiles_ci, tci = b - tco * p.tiles_ci;
  const int co0 = tco * 64, ci0 = tci * 64;
  const int NSTEP = p.steps_per_split;                    // multiple of 6
  const long q_begin = (long)split * NSTEP * 16;          // first dy position of this range
  const int ring_rows = p.ring_steps * 16;
  unsigned char* const xring = dsm;                                        // (ring_rows + 16) rows
  unsigned char* const dyring = dsm + (size_t)(ring_rows + 16) * WS_ROW;   // 2 x 16 rows

  f32x16 acc[9];

  if (wave >= 4) {
    // ------------------------------------------------------------------ producers (256 lanes: 16 rows x 16 channel quads)
    const int prow = (tid - 256) >> 4, cq = tid & 15;
    const bool x_cok = ci0 + cq * 4 < p.ci_read, d_cok = co0 + cq * 4 < p.co_read;
    auto decode = [&](long q) {
      Pos s;
      s.q = q;
      const long qq = q < 0 ? 0 : q;
      const long per = (long)p.Hd * p.Wd;
      s.n = (int)(qq / per);
      const int rem = (int)(qq - (long)s.n * per);
      s.yp = rem / p.Wd;
      s.xp = rem - s.yp * p.Wd;
      return s;
    };
    auto advance = [&](Pos& s) {   // += 16 positions; Wd > 16: at most one row carry
      s.q += 16;
      if (s.q < 16) {              // the stream started before position 0 and is (re)entering the tensor
        if (s.q >= 0) s = decode(s.q);
        return;
      }
      s.xp += 16;
      if (s.xp >= p.Wd) {
        s.xp -= p.Wd;
        if (++s.yp >= p.Hd) {
          s.yp = 0;
          ++s.n;
        }
      }
    };
    f32x4 xsc = {0.f, 0.f, 0.f, 0.f}, xsh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (XAFF) {
      xsc = *reinterpret_cast<const f32x4*>(x_cok ? p.x_scale + ci0 + cq * 4 : p.zero);
      xsh = *reinterpret_cast<const f32x4*>(x_cok ? p.x_shift + ci0 + cq * 4 : p.zero);
    }
    auto fetch = [&](const Pos& s, const float* base, int ld, int c0, bool cok, bool transform = false) {
      const bool real = s.q >= 0 && s.q < p.Q && s.yp < p.H && s.xp < p.W && cok;
      const long pix = ((long)s.n * p.H + s.yp) * p.W + s.xp;
      if constexpr (IO16) {
        const bf16_t* src = real ? reinterpret_cast<const bf16_t*>(base) + pix * ld + c0 + cq * 4 : reinterpret_cast<const bf16_t*>(p.zero);
        const u32x2 raw = *reinterpret_cast<const u32x2*>(src);
        return f32x4{__uint_as_float(raw[0]), __uint_as_float(raw[1]), 0.f, 0.f};
      }
      const float* src = real ? base + pix * ld + c0 + cq * 4 : p.zero;
      f32x4 v = *reinterpret_cast<const f32x4*>(src);
      if constexpr (XAFF) {
        if (transform) {
          v = affine_relu4(v, xsc, xsh);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = real ? v[e] : 0.f;
        }
      }
      return v;
    };
    auto conv_write = [&](const f32x4 v, unsigned char* row) {
      if constexpr (IO16) {
        *reinterpret_cast<u32x2*>(row + cq * 8) = u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])};
        return;
      }
      u32x2 hi, lo;
      unsigned h, l;
      split_pair<PREC>(v[0], v[1], h, l); hi[0] = h; lo[0] = l;
      split_pair<PREC>(v[2], v[3], h, l); hi[1] = h; lo[1] = l;
      *reinterpret_cast<u32x2*>(row + cq * 8) = hi;
      if (PREC == 3) *reinterpret_cast<u32x2*>(row + 128 + cq * 8) = lo;
    };
    // x step k: positions q_begin - RL + 16 k + prow -> ring slot k % ring_steps; dy step i: positions q_begin + 16 i + prow
    Pos xs = decode(q_begin - p.RL + prow), ds = decode(q_begin + prow);
    int xslot = 0;   // ring slot of the next x step to be WRITTEN
    auto write_x = [&](const f32x4 v) {
      unsigned char* row = xring + (size_t)(xslot * 16 + prow) * WS_ROW;
      conv_write(v, row);
      if (xslot == 0) conv_write(v, row + (size_t)ring_rows * WS_ROW);   // mirror of the first 16 rows: windows never wrap
      xslot = xslot + 1 == p.ring_steps ? 0 : xslot + 1;
    };
    auto load_x = [&]() {
      const f32x4 v = fetch(xs, p.x, p.ldx, ci0, x_cok, true);
      advance(xs);
      return v;
    };
    auto load_d = [&]() {
      const f32x4 v = fetch(ds, p.dy, p.lddy, co0, d_cok);
      advance(ds);
      return v;
    };
    f32x4 xbuf[3], dbuf[3];
    // ---- prologue: x steps 0 .. lead and dy step 0 into LDS (lead + 1 is a multiple of 3: three loads in flight per trip)
    for (int k = 0; k <= p.lead; k += 3) {
#pragma unroll
      for (int s = 0; s < 3; ++s) xbuf[s] = load_x();
#pragma unroll
      for (int s = 0; s < 3; ++s) write_x(xbuf[s]);
    }
    dbuf[0] = load_d();
    conv_write(dbuf[0], dyring + (size_t)prow * WS_ROW);
    // requested and in flight from here on: x steps lead+1 .. lead+3 (sets 0, 1, 2) and dy steps 1 .. 3 (sets 1, 2, 0)
#pragma unroll
    for (int s = 0; s < 3; ++s) xbuf[s] = load_x();
    dbuf[1] = load_d();
    dbuf[2] = load_d();
    dbuf[0] = load_d();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // B_0
    // ---- interval i (consumers multiply K step i): write x step i + lead + 1 and dy step i + 1, request the steps 3 later
    for (int i0 = 0; i0 < NSTEP; i0 += 3) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        write_x(xbuf[r]);
        xbuf[r] = load_x();
        conv_write(dbuf[(r + 1) % 3], dyring + (size_t)(((i0 + r + 1) & 1) * 16 + prow) * WS_ROW);
        dbuf[(r + 1) % 3] = load_d();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // B_{i+1}
      }
    }
  } else {
    // ------------------------------------------------------------------ consumers: transposing LDS reads + MFMA
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int bi = wave >> 1, bj = wave & 1;
    // lane -> (position row within the 16-position step, channel quad) of its transposing reads: 16-lane group g covers
    // channels 16 (g & 1) .. +15 and positions 8 (g >> 1) .. +7; inside the group source lane l reads position (l >> 2)
    // (+4 for the second read), channels 4 (l & 3) .. +3
    const int g = lane >> 4, l16 = lane & 15;
    const unsigned lane_row = (unsigned)(8 * (g >> 1) + (l16 >> 2));
    const unsigned lane_ch = (unsigned)(16 * (g & 1) + 4 * (l16 & 3));
    const unsigned a_off = lane_row * WS_ROW + (unsigned)(32 * bi + lane_ch) * 2;
    const unsigned b_off = lane_row * WS_ROW + (unsigned)(32 * bj + lane_ch) * 2;
    auto tr_pair = [&](const unsigned char* base, unsigned off) {
      const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off));
      const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off + 4 * WS_ROW));
      return bf16x8{u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
    };
    bf16x8 a_hi[2], a_lo[2];          // dy fragments of this K step / the next one
    bf16x8 x_hi[3], x_lo[3];          // x fragments: tap t lives in set t % 3
    // window of tap t at K step i: ring rows (16 i + RL + toff[t]) mod ring_rows .. +15 (the mirror makes the wrap harmless)
    int wbase = p.RL;                 // 16 i + RL, kept modulo ring_rows
    auto window = [&](int t) {
      int r = wbase + p.toff[t];
      r = r >= ring_rows ? r - ring_rows : (r < 0 ? r + ring_rows : r);
      return xring + (size_t)r * WS_ROW;
    };
    auto read_x = [&](int t) {
      const unsigned char* w = window(t);
      x_hi[t % 3] = tr_pair(w, b_off);
      if (PREC == 3) x_lo[t % 3] = tr_pair(w, b_off + 128);
    };
    auto read_a = [&](int set, int slot) {
      const unsigned char* w = dyring + (size_t)slot * 16 * WS_ROW;
      a_hi[set] = tr_pair(w, a_off);
      if (PREC == 3) a_lo[set] = tr_pair(w, a_off + 128);
    };
    // One K step: nine taps of 3 (bf16x3) or 1 MFMAs on this wave's 32 x 32 block.  The x fragment of tap t + 2 is read in
    // the shadow of tap t; the step barrier (next dy rows / next x rows written) sits before tap 7, after which the next
    // step's dy fragment and its first two x fragments are fetched.
    auto step = [&](auto setc) {
      constexpr int SET = decltype(setc)::value;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t == 7) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          wbase += 16;
          wbase = wbase >= ring_rows ? wbase - ring_rows : wbase;
        }
        const bf16x8 xh = x_hi[t % 3], xl = x_lo[t % 3];
        if (PREC == 3) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[SET], xh, acc[t], 0, 0, 0);
          if (t == 7) read_a(SET ^ 1, SET ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[SET], xl, acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        } else if (t == 7) {
          read_a(SET ^ 1, SET ^ 1);
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[SET], xh, acc[t], 0, 0, 0);
        // tap t + 2 of this step (t <= 6), or taps 0 / 1 of the next step (t = 7, 8: wbase already points there)
        read_x((t + 2) % 9);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    __builtin_amdgcn_s_barrier();   // B_0
    asm volatile("" ::: "memory");
    read_a(0, 0);
    read_x(0);
    read_x(1);
    __builtin_amdgcn_sched_barrier(0);
    for (int i = 0; i < NSTEP; i += 2) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
    }
    // ---- store this range's partial tile: out[split][co][tap][ci] (a lane owns one ci column of 16 co rows per tap)
    float* out = p.out + (size_t)split * p.slab;
    const int ci = ci0 + 32 * bj + (lane & 31);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.co_write && ci < p.ci_write) out[(size_t)co * p.ldw + (size_t)t * p.cin_w + ci] = acc[t][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pointwise (1x1, stride 1) weight gradient with the same division of labour:  dw[co, ci] = sum_p dy[p, co] * x[p, ci].
//
// conv_wgrad.hip's LDS-DMA kernel stages raw fp32 and lets every MFMA wave split its own fragments (6 VALU per MFMA, MFMA
// pipe 18.5 % busy over the 1x1 layers).  Here four producer waves load a 32-position K step of both operands (plain loads,
// three steps ahead), split to bf16 hi/lo once and write [position][channel] rows (the 320-byte rows of the strip kernel,
// one array of rows per 64 channels); four consumer waves own (32 NCO) x (32 NCI) blocks of a (64 NCO) x (64 NCI) tile and
// fetch position-major fragments with ds_read_b64_tr_b16.  Per 16-position sub-step a consumer issues 3 NCO NCI MFMAs
// against 4 (NCO + NCI) transposing reads; the producers' VALU work per K step, (NCO + NCI) * 64 * 32 / 256 values per
// lane, runs on the VALU pipe of the same SIMDs while the MFMA pipe multiplies.  One barrier per K step (two stages of LDS).
// Replaces convolution_backward(weight) of the stride-1 1x1 nn.Conv2d of resnet.py:18-26,33-53, aspp.py:11-19,86-88.
struct PwArgs {
  const float* dy;
  const float* x;
  float* out;          // slab 0 (or dw itself when splitk == 1)
  const float* zero;
  long M;              // positions (N * H * W)
  int co_read, co_write, ci_read, ci_write;
  int lddy, ldx, ldw;
  int steps_per_split; // 32-position K steps per range, multiple of 6
  int tiles_co, tiles_ci;
  long slab;
  const float* x_scale;   // optional x' = max(x * x_scale[c] + x_shift[c], 0) (as StripArgs)
  const float* x_shift;
};

constexpr int PW_ARRAY = 32 * WS_ROW;   // one 64-channel array of a stage: 32 positions

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int PREC, int NCO, int NCI, bool XAFF = false, bool IO16 = false>   // IO16: as conv_wgrad_strip_kernel
__global__ __launch_bounds__(512) void conv_wgrad_pw_kernel(const PwArgs p) {
  static_assert(!IO16 || (PREC == 1 && !XAFF), "bf16-stored operands: plain bf16 products, no producer-side transform");
  constexpr int ESZ = IO16 ? 2 : 4;   // bytes per stored element
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  constexpr int NARR = NCO + NCI;
  constexpr int STAGE = NARR * PW_ARRAY;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int ntiles = p.tiles_co * p.tiles_ci;
  const int split = b / ntiles;
  b -= split * ntiles;
  const int tco = b / p.tiles_ci, tci = b - tco * p.tiles_ci;
  const int co0 = tco * 64 * NCO, ci0 = tci * 64 * NCI;
  const int NSTEP = p.steps_per_split;
  const long q_begin = (long)split * NSTEP * 32;

  if (wave >= 4) {
    // ------------------------------------------------------------------ producers (256 lanes: 16 rows x 16 channel quads)
    const int prow = (tid - 256) >> 4, cq = tid & 15;
    long q = q_begin + prow;    // position of this lane's next load (advances by 16 per half step)
    bool cok[NARR];
    const unsigned char* ptr[NARR];
    long ld16[NARR];
#pragma unroll
    for (int a = 0; a < NARR; ++a) {
      const bool isd = a < NCO;
      const int c = (isd ? co0 + 64 * a : ci0 + 64 * (a - NCO)) + cq * 4;
      const long ld = isd ? p.lddy : p.ldx;
      cok[a] = c < (isd ? p.co_read : p.ci_read);
      ptr[a] = reinterpret_cast<const unsigned char*>(isd ? p.dy : p.x) + (c + q * ld) * ESZ;
      ld16[a] = 16 * ld * ESZ;
    }
    f32x4 xsc[NCI], xsh[NCI];      // (XAFF) scale / shift of this lane's four channels in each x array
    if constexpr (XAFF) {
#pragma unroll
      for (int a = 0; a < NCI; ++a) {
        const int c = ci0 + 64 * a + cq * 4;
        xsc[a] = *reinterpret_cast<const f32x4*>(cok[NCO + a] ? p.x_scale + c : p.zero);
        xsh[a] = *reinterpret_cast<const f32x4*>(cok[NCO + a] ? p.x_shift + c : p.zero);
      }
    }
    f32x4 buf[3][2 * NARR];
    auto load_step = [&](f32x4* dst) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool rok = q < p.M;
#pragma unroll
        for (int a = 0; a < NARR; ++a) {
          const unsigned char* src = rok && cok[a] ? ptr[a] : reinterpret_cast<const unsigned char*>(p.zero);
          if constexpr (IO16) {
            const u32x2 raw = *reinterpret_cast<const u32x2*>(src);
            dst[h * NARR + a] = f32x4{__uint_as_float(raw[0]), __uint_as_float(raw[1]), 0.f, 0.f};
          } else {
            dst[h * NARR + a] = *reinterpret_cast<const f32x4*>(src);
          }
          ptr[a] += ld16[a];
        }
        q += 16;
      }
    };
    auto conv_write = [&](const f32x4 v, unsigned char* row) {
      if constexpr (IO16) {
        *reinterpret_cast<u32x2*>(row + cq * 8) = u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])};
        return;
      }
      u32x2 hi, lo;
      unsigned h, l;
      split_pair<PREC>(v[0], v[1], h, l); hi[0] = h; lo[0] = l;
      split_pair<PREC>(v[2], v[3], h, l); hi[1] = h; lo[1] = l;
      *reinterpret_cast<u32x2*>(row + cq * 8) = hi;
      if (PREC == 3) *reinterpret_cast<u32x2*>(row + 128 + cq * 8) = lo;
    };
    auto write_step = [&](const f32x4* src, int stage) {
      unsigned char* s0 = dsm + (size_t)stage * STAGE + (size_t)prow * WS_ROW;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int a = 0; a < NARR; ++a) {
          f32x4 v = src[h * NARR + a];
          if constexpr (XAFF) {
            if (a >= NCO) {
              v = affine_relu4(v, xsc[a - NCO], xsh[a - NCO]);   // (rows past M become relu(shift): they multiply dy rows that are zero)
            }
          }
          conv_write(v, s0 + (size_t)a * PW_ARRAY + (size_t)h * 16 * WS_ROW);
        }
      }
    };
    // prologue: step 0 into stage 0; steps 1, 2, 3 requested into register sets 1, 2, 0
    load_step(buf[0]);
    write_step(buf[0], 0);
    load_step(buf[1]);
    load_step(buf[2]);
    load_step(buf[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // B_0
    // interval i (consumers multiply step i): write step i + 1 into the other stage, request step i + 4
    for (int i0 = 0; i0 < NSTEP; i0 += 3) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        write_step(buf[(r + 1) % 3], (i0 + r + 1) & 1);
        load_step(buf[(r + 1) % 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // B_{i+1}
      }
    }
  } else {
    // ------------------------------------------------------------------ consumers: transposing LDS reads + MFMA
    f32x16 acc[NCO][NCI];
#pragma unroll
    for (int m = 0; m < NCO; ++m)
#pragma unroll
      for (int n = 0; n < NCI; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int bi = wave >> 1, bj = wave & 1;
    const int cob = bi * 32 * NCO, cib = bj * 32 * NCI;       // this wave's channel offsets inside the tile
    const int g = lane >> 4, l16 = lane & 15;
    const unsigned lane_row = (unsigned)(8 * (g >> 1) + (l16 >> 2));
    const unsigned lane_ch = (unsigned)(16 * (g & 1) + 4 * (l16 & 3));
    // byte offset of block m's hi fragment inside a stage (sub-step 0): array, row, channel
    unsigned a_off[NCO], b_off[NCI];
#pragma unroll
    for (int m = 0; m < NCO; ++m) {
      const int c = cob + 32 * m;
      a_off[m] = (unsigned)(c / 64) * PW_ARRAY + lane_row * WS_ROW + (unsigned)(c % 64 + lane_ch) * 2;
    }
#pragma unroll
    for (int n = 0; n < NCI; ++n) {
      const int c = cib + 32 * n;
      b_off[n] = (unsigned)(NCO + c / 64) * PW_ARRAY + lane_row * WS_ROW + (unsigned)(c % 64 + lane_ch) * 2;
    }
    auto tr_pair = [&](const unsigned char* base, unsigned off) {
      const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off));
      const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off + 4 * WS_ROW));
      return bf16x8{u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
    };
    constexpr int NL = PREC == 3 ? 2 : 1;             // hi (and lo) planes
    constexpr int NREAD = NL * NARR;                  // fragment reads per sub-step
    bf16x8 fa[2][NCO][2], fb[2][NCI][2];              // [register set][block][hi, lo]
    // read number k of a sub-step: dy blocks first (hi, lo), then x blocks
    auto read_k = [&](auto setc, auto kc, const unsigned char* rows) {
      constexpr int SET = decltype(setc)::value, K = decltype(kc)::value;
      if constexpr (K < NREAD) {
        constexpr int blk = K / NL, pl = K % NL;
        if constexpr (blk < NCO)
          fa[SET][blk][pl] = tr_pair(rows, a_off[blk] + 128 * pl);
        else
          fb[SET][blk - NCO][pl] = tr_pair(rows, b_off[blk - NCO] + 128 * pl);
      }
    };
    // one 16-position sub-step on register set SET; the other set is filled from `rows` in the shadow of the MFMAs
    auto substep = [&](auto setc, const unsigned char* rows) {
      constexpr int SET = decltype(setc)::value;
      constexpr int NMF = NCO * NCI * (PREC == 3 ? 3 : 1);
      constexpr int PER = (NREAD + NMF - 1) / NMF;    // reads issued after each MFMA
      auto reads_after = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        read_k(std::integral_constant<int, SET ^ 1>{}, std::integral_constant<int, J * PER>{}, rows);
        if constexpr (PER > 1) read_k(std::integral_constant<int, SET ^ 1>{}, std::integral_constant<int, J * PER + 1>{}, rows);
        if constexpr (PER > 2) read_k(std::integral_constant<int, SET ^ 1>{}, std::integral_constant<int, J * PER + 2>{}, rows);
        if constexpr (PER > 3) read_k(std::integral_constant<int, SET ^ 1>{}, std::integral_constant<int, J * PER + 3>{}, rows);
        __builtin_amdgcn_sched_barrier(0);
      };
      // product pl of block (m, n); consecutive MFMAs go to different accumulators (lo*hi, hi*lo, hi*hi over all blocks)
      auto prod = [&](auto plc, auto mc, auto nc) {
        constexpr int pl = decltype(plc)::value, m = decltype(mc)::value, n = decltype(nc)::value;
        constexpr int j = pl * NCO * NCI + m * NCI + n;
        constexpr int ia = PREC == 3 ? (pl == 0 ? 1 : 0) : 0, ib = PREC == 3 ? (pl == 1 ? 1 : 0) : 0;
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][m][ia], fb[SET][n][ib], acc[m][n], 0, 0, 0);
        reads_after(std::integral_constant<int, j>{});
      };
      auto blocks = [&](auto plc) {
        static_for<NCO * NCI>([&](auto bc) {
          constexpr int B = decltype(bc)::value;
          prod(plc, std::integral_constant<int, B / NCI>{}, std::integral_constant<int, B % NCI>{});
        });
      };
      blocks(std::integral_constant<int, 0>{});
      if constexpr (PREC == 3) {
        blocks(std::integral_constant<int, 1>{});
        blocks(std::integral_constant<int, 2>{});
      }
    };
    __builtin_amdgcn_s_barrier();   // B_0
    asm volatile("" ::: "memory");
    {
      // sub-step 0 of step 0 into register set 0
      const unsigned char* rows = dsm;
#pragma unroll
      for (int m = 0; m < NCO; ++m)
#pragma unroll
        for (int pl = 0; pl < NL; ++pl) fa[0][m][pl] = tr_pair(rows, a_off[m] + 128 * pl);
#pragma unroll
      for (int n = 0; n < NCI; ++n)
#pragma unroll
        for (int pl = 0; pl < NL; ++pl) fb[0][n][pl] = tr_pair(rows, b_off[n] + 128 * pl);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int i = 0; i < NSTEP; ++i) {
      const unsigned char* cur = dsm + (size_t)(i & 1) * STAGE;
      const unsigned char* nxt = dsm + (size_t)((i + 1) & 1) * STAGE;
      substep(std::integral_constant<int, 0>{}, cur + 16 * WS_ROW);   // multiply positions 0..15, fetch 16..31
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                   // B_{i+1}: the other stage holds step i + 1
      asm volatile("" ::: "memory");
      substep(std::integral_constant<int, 1>{}, nxt);                 // multiply positions 16..31, fetch step i + 1's 0..15
    }
    // ---- store this range's partial tile: out[split][co][ci]
    float* out = p.out + (size_t)split * p.slab;
#pragma unroll
    for (int m = 0; m < NCO; ++m)
#pragma unroll
      for (int n = 0; n < NCI; ++n) {
        const int ci = ci0 + cib + 32 * n + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + cob + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (co < p.co_write && ci < p.ci_write) out[(size_t)co * p.ldw + ci] = acc[m][n][r];
        }
      }
  }
}

__global__ __launch_bounds__(256) void strip_reduce_kernel(const f32x4* __restrict__ part, f32x4* __restrict__ dw, long n4, int splitk,
                                                          long slab4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 s = part[i];
#pragma unroll 4
    for (int k = 1; k < splitk; ++k) s += part[(size_t)k * slab4 + i];
    dw[i] = s;
  }
}

// (dw may be a slice of a gradient bucket at any 4-byte offset -- parallel.GradSync's zero-copy hand-off: scalar form)
__global__ __launch_bounds__(256) void strip_reduce1_kernel(const float* __restrict__ part, float* __restrict__ dw, long n, int splitk,
                                                           long slab) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = part[i];
#pragma unroll 4
    for (int k = 1; k < splitk; ++k) s += part[(size_t)k * slab + i];
    dw[i] = s;
  }
}

struct StripPlan {
  int ok, splitk, steps_per_split, RL, lead, ring_steps, lds_bytes;
  long Q, workspace_floats;
};

StripPlan strip_plan(int N, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad_h, int pad_w, int dil, int co,
                     int ci) {
  StripPlan s{};
  if (KH != 3 || KW != 3 || stride != 1 || H != Ho || W != Wo || pad_h != dil || pad_w != dil) return s;
  if (co < 64 || ci < 64) return s;
  const int Wd = W + dil, Hd = H + dil;
  if (Wd <= 16) return s;
  const int R = dil * (Wd + 1);
  int RL = (R + 15) / 16 * 16;
  int lead = (RL + R + 15) >> 4;
  while ((lead + 1) % 3) {      // producer register sets are indexed by (K step) % 3
    RL += 16;
    lead = (RL + R + 15) >> 4;
  }
  s.RL = RL;
  s.lead = lead;
  s.ring_steps = lead + 2;
  s.lds_bytes = ((s.ring_steps + 1) * 16 + 2 * 16) * WS_ROW;
  if (s.lds_bytes > 160 * 1024) return s;
  s.Q = (long)N * Hd * Wd;
  const long steps = (s.Q + 15) / 16;
  const int tiles = ((co + 63) / 64) * ((ci + 63) / 64);
  // split-K: aim at ZS3_WGRAD_STRIP_WGS workgroups (default 192), at least 36 K steps per range (the prologue fills lead + 1
  // ring steps).  Same-box sweep inside the training step (ms per step, tools/probe/ab_env.sh): 128 -> 46.11, 192 -> 46.05 /
  // 46.12, 256 -> 46.10 / 46.15, 384 -> 46.67 / 46.46, 512 -> 46.83 / 46.59, 1024 -> 46.85 / 46.92 (without this kernel: 47.7):
  // the launches run next to the dgrad chain, and every split costs a [Cout][9][Cin] slab of HBM traffic.
  static const int wgs = getenv("ZS3_WGRAD_STRIP_WGS") ? atoi(getenv("ZS3_WGRAD_STRIP_WGS")) : 192;
  long want = (wgs + tiles - 1) / tiles;
  long maxs = steps / 36;
  if (maxs < 1) maxs = 1;
  long sk = want < maxs ? want : maxs;
  if (sk > 256) sk = 256;
  long per = (steps + sk - 1) / sk;
  per = (per + 5) / 6 * 6;
  sk = (steps + per - 1) / per;
  s.splitk = (int)sk;
  s.steps_per_split = (int)per;
  s.workspace_floats = sk > 1 ? sk * (long)co * 9 * ci : 0;
  s.ok = 1;
  return s;
}

struct PwPlan {
  int ok, nco, nci, splitk, steps_per_split, lds_bytes;
  long workspace_floats;
};

PwPlan pw_plan(long M, int co, int ci, bool prec3 = false) {
  PwPlan s{};
  if (co < 64 || ci < 64 || M < 32 * 12) return s;
  s.nco = co >= 128 ? 2 : 1;
  s.nci = ci >= 128 ? 2 : 1;
  // wide tiles for the layers whose larger side has >= 1024 channels (x3 products only): a (4, 2) or (2, 4) consumer block reads
  // 6 arrays per K step for 8 products instead of 4 for 4 -- 25 % fewer operand bytes through L2 and the fabric, 33 % fewer LDS
  // reads per MFMA
  // (alone, B = 16 at 33^2: 256 <-> 1024 channels 60.0 -> 56.7 us, 512 <-> 2048 192 -> 170 us; inside the step the change is not
  // visible: 43.04 / 43.04 / 42.86 against 43.19 / 43.05 / 42.81 ms, tools/probe/ab_quick.sh.  ZS3_WGRAD_PW_WIDE=0 turns it off)
  static const int wide = getenv("ZS3_WGRAD_PW_WIDE") ? atoi(getenv("ZS3_WGRAD_PW_WIDE")) : 1024;
  if (wide && prec3 && co >= 256 && ci >= 256 && (co >= wide || ci >= wide)) {
    if (co >= ci) s.nco = 4; else s.nci = 4;
  }
  s.lds_bytes = 2 * (s.nco + s.nci) * PW_ARRAY;
  const long steps = (M + 31) / 32;
  const int tiles = ((co + 64 * s.nco - 1) / (64 * s.nco)) * ((ci + 64 * s.nci - 1) / (64 * s.nci));
  // split-K: aim at ZS3_WGRAD_PW_WGS workgroups (512 threads, up to 80 KB of LDS: one per CU), at least 12 K steps per range.
  // Inside the training step (same box, ms per step, tools/probe/ab_env.sh): 128 -> 46.12 / 46.06, 160 -> 46.43 / 46.37,
  // 192 -> 47.59 / 47.56, 224 -> 47.55 / 47.52, 256 -> 46.35 / 46.16; alone 256 is the fastest (the whole pass 12.9 ms).
  static const int wgs = getenv("ZS3_WGRAD_PW_WGS") ? atoi(getenv("ZS3_WGRAD_PW_WGS")) : 128;
  long want = (wgs + tiles - 1) / tiles;
  long maxs = steps / 12;
  if (maxs < 1) maxs = 1;
  long sk = want < maxs ? want : maxs;
  if (sk > 256) sk = 256;
  long per = (steps + sk - 1) / sk;
  per = (per + 5) / 6 * 6;
  sk = (steps + per - 1) / per;
  s.splitk = (int)sk;
  s.steps_per_split = (int)per;
  s.workspace_floats = sk > 1 ? sk * (long)co * ci : 0;
  s.ok = 1;
  return s;
}

template <int PREC, int NCO, int NCI, bool XAFF, bool IO16 = false>
int launch_pw_x(const PwArgs& a, int grid, int lds, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_pw_kernel<PREC, NCO, NCI, XAFF, IO16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return -4;
    configured = true;
  }
  hipLaunchKernelGGL((conv_wgrad_pw_kernel<PREC, NCO, NCI, XAFF, IO16>), dim3(grid), dim3(512), lds, st, a);
  return ZS3_LAUNCH_CHECK();
}
template <int PREC, int NCO, int NCI>
int launch_pw(const PwArgs& a, int grid, int lds, int io16, hipStream_t st) {
  if constexpr (PREC == 1) {
    if (io16) return launch_pw_x<1, NCO, NCI, false, true>(a, grid, lds, st);
  }
  return a.x_scale ? launch_pw_x<PREC, NCO, NCI, true>(a, grid, lds, st) : launch_pw_x<PREC, NCO, NCI, false>(a, grid, lds, st);
}

int reduce_slabs(const float* workspace, float* dw, long n, int splitk, hipStream_t st) {
  if (((uintptr_t)dw & 15) == 0 && n % 4 == 0) {
    const long n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(strip_reduce_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f32x4*>(workspace),
                       reinterpret_cast<f32x4*>(dw), n4, splitk, n4);
  } else {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(strip_reduce1_kernel, dim3(blocks), dim3(256), 0, st, workspace, dw, n, splitk, n);
  }
  return ZS3_LAUNCH_CHECK();
}

}  // namespace

// Pointwise (1x1, stride 1, no padding) weight gradient: dw[co, ci] = sum over the M = N*H*W positions of dy[p, co] x[p, ci].
// zs3_conv_wgrad_pw_plan returns 1 when the layer is eligible (>= 64 channels on both sides) with the split-K factor and the
// workspace floats the launch needs, else 0 (use zs3_conv_wgrad).
extern "C" int zs3_conv_wgrad_pw_plan(long M, int co, int ci, int* splitk_out, long* workspace_floats) {
  const PwPlan s = pw_plan(M, co, ci), w = pw_plan(M, co, ci, true);     // (the caller does not say which products: the larger workspace)
  if (splitk_out) *splitk_out = s.splitk > w.splitk ? s.splitk : w.splitk;
  if (workspace_floats) *workspace_floats = s.workspace_floats > w.workspace_floats ? s.workspace_floats : w.workspace_floats;
  return s.ok;
}

extern "C" int zs3_conv_wgrad_pw(const float* dy, const float* x, float* dw, float* workspace, long M, int co_read, int co_write,
                                 int ci_read, int ci_write, int lddy, int ldx, int prec, const void* zero_page,
                                 const float* x_scale, const float* x_shift, int io, void* stream) {
  if (co_read % 4 || ci_read % 4 || lddy % 4 || ldx % 4 || (prec != 1 && prec != 3) || zero_page == nullptr) return -1;
  if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)zero_page & 15) || ((uintptr_t)dw & 3)) return -2;
  if ((x_scale == nullptr) != (x_shift == nullptr) || ((uintptr_t)x_scale & 15) || ((uintptr_t)x_shift & 15)) return -1;
  if (io != 0 && io != 3) return -7;                 // one of the two operands bf16: zs3_conv_wgrad
  if (io && (prec != 1 || x_scale)) return -7;       // bf16-stored operands: plain-bf16 products, no transform
  const int io16 = io ? 1 : 0;
  const PwPlan s = pw_plan(M, co_write, ci_write, prec == 3 && !io);
  if (!s.ok) return -7;
  if (s.splitk > 1 && (workspace == nullptr || ((uintptr_t)workspace & 15))) return -3;
  PwArgs a{};
  a.dy = dy; a.x = x; a.zero = (const float*)zero_page;
  a.x_scale = x_scale; a.x_shift = x_shift;
  a.M = M;
  a.co_read = co_read; a.co_write = co_write; a.ci_read = ci_read; a.ci_write = ci_write;
  a.lddy = lddy; a.ldx = ldx; a.ldw = ci_write;
  a.steps_per_split = s.steps_per_split;
  a.tiles_co = (co_write + 64 * s.nco - 1) / (64 * s.nco);
  a.tiles_ci = (ci_write + 64 * s.nci - 1) / (64 * s.nci);
  a.slab = (long)co_write * ci_write;
  a.out = s.splitk > 1 ? workspace : dw;
  hipStream_t st = (hipStream_t)stream;
  const int grid = a.tiles_co * a.tiles_ci * s.splitk;
  int rc;
  const int key = s.nco == 4 ? 8 : s.nci == 4 ? 9 : (prec == 3 ? 4 : 0) + (s.nco == 2 ? 2 : 0) + (s.nci == 2 ? 1 : 0);
  switch (key) {
    case 8: rc = launch_pw<3, 4, 2>(a, grid, s.lds_bytes, io16, st); break;
    case 9: rc = launch_pw<3, 2, 4>(a, grid, s.lds_bytes, io16, st); break;
    case 0: rc = launch_pw<1, 1, 1>(a, grid, s.lds_bytes, io16, st); break;
    case 1: rc = launch_pw<1, 1, 2>(a, grid, s.lds_bytes, io16, st); break;
    case 2: rc = launch_pw<1, 2, 1>(a, grid, s.lds_bytes, io16, st); break;
    case 3: rc = launch_pw<1, 2, 2>(a, grid, s.lds_bytes, io16, st); break;
    case 4: rc = launch_pw<3, 1, 1>(a, grid, s.lds_bytes, io16, st); break;
    case 5: rc = launch_pw<3, 1, 2>(a, grid, s.lds_bytes, io16, st); break;
    case 6: rc = launch_pw<3, 2, 1>(a, grid, s.lds_bytes, io16, st); break;
    default: rc = launch_pw<3, 2, 2>(a, grid, s.lds_bytes, io16, st); break;
  }
  if (rc) return rc;
  if (s.splitk > 1) rc = reduce_slabs(workspace, dw, a.slab, s.splitk, st);
  return rc;
}

// Eligibility + split-K plan of the strip-resident weight-gradient kernel: returns 1 and the workspace size (floats) when
// zs3_conv_wgrad_strip can run the layer (3x3, stride 1, same size, pad = dilation, >= 64 channels on both sides, ring fits
// the LDS), else 0 (use zs3_conv_wgrad).
extern "C" int zs3_conv_wgrad_strip_plan(int N, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad_h, int pad_w,
                                         int dil, int co, int ci, int* splitk_out, long* workspace_floats) {
  const StripPlan s = strip_plan(N, H, W, Ho, Wo, KH, KW, stride, pad_h, pad_w, dil, co, ci);
  if (splitk_out) *splitk_out = s.splitk;
  if (workspace_floats) *workspace_floats = s.workspace_floats;
  return s.ok;
}

extern "C" int zs3_conv_wgrad_strip(const float* dy, const float* x, float* dw, float* workspace, int N, int H, int W, int dil,
                                    int co_read, int co_write, int ci_read, int ci_write, int lddy, int ldx, int prec,
                                    const void* zero_page, const float* x_scale, const float* x_shift, int io, void* stream) {
  if (co_read % 4 || ci_read % 4 || lddy % 4 || ldx % 4 || (prec != 1 && prec != 3) || zero_page == nullptr) return -1;
  if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)zero_page & 15) || ((uintptr_t)dw & 3)) return -2;
  if ((x_scale == nullptr) != (x_shift == nullptr) || ((uintptr_t)x_scale & 15) || ((uintptr_t)x_shift & 15)) return -1;
  if (io != 0 && io != 3) return -7;
  if (io && (prec != 1 || x_scale)) return -7;
  const StripPlan s = strip_plan(N, H, W, H, W, 3, 3, 1, dil, dil, dil, co_write, ci_write);
  if (!s.ok) return -7;
  if (s.splitk > 1 && (workspace == nullptr || ((uintptr_t)workspace & 15))) return -3;
  StripArgs a{};
  a.dy = dy; a.x = x; a.zero = (const float*)zero_page;
  a.x_scale = x_scale; a.x_shift = x_shift;
  a.N = N; a.H = H; a.W = W;
  a.co_read = co_read; a.co_write = co_write; a.ci_read = ci_read; a.ci_write = ci_write;
  a.lddy = lddy; a.ldx = ldx; a.cin_w = ci_write; a.ldw = 9 * ci_write;
  a.Hd = H + dil; a.Wd = W + dil; a.Q = s.Q;
  a.RL = s.RL; a.lead = s.lead; a.ring_steps = s.ring_steps; a.steps_per_split = s.steps_per_split;
  a.tiles_co = (co_write + 63) / 64; a.tiles_ci = (ci_write + 63) / 64;
  a.slab = (long)co_write * a.ldw;
  a.out = s.splitk > 1 ? workspace : dw;
  for (int t = 0; t < 9; ++t) a.toff[t] = (t / 3 - 1) * dil * a.Wd + (t % 3 - 1) * dil;
  hipStream_t st = (hipStream_t)stream;
  const int grid = a.tiles_co * a.tiles_ci * s.splitk;
  static bool configured[5] = {false, false, false, false, false};
  const int pi = io ? 4 : (prec == 1 ? 0 : 1) + (x_scale ? 2 : 0);
  const void* fns[5] = {reinterpret_cast<const void*>(&conv_wgrad_strip_kernel<1, false>),
                        reinterpret_cast<const void*>(&conv_wgrad_strip_kernel<3, false>),
                        reinterpret_cast<const void*>(&conv_wgrad_strip_kernel<1, true>),
                        reinterpret_cast<const void*>(&conv_wgrad_strip_kernel<3, true>),
                        reinterpret_cast<const void*>(&conv_wgrad_strip_kernel<1, false, true>)};
  if (!configured[pi]) {
    if (hipFuncSetAttribute(fns[pi], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -4;
    configured[pi] = true;
  }
  switch (pi) {
    case 0: hipLaunchKernelGGL((conv_wgrad_strip_kernel<1, false>), dim3(grid), dim3(512), s.lds_bytes, st, a); break;
    case 1: hipLaunchKernelGGL((conv_wgrad_strip_kernel<3, false>), dim3(grid), dim3(512), s.lds_bytes, st, a); break;
    case 2: hipLaunchKernelGGL((conv_wgrad_strip_kernel<1, true>), dim3(grid), dim3(512), s.lds_bytes, st, a); break;
    case 4: hipLaunchKernelGGL((conv_wgrad_strip_kernel<1, false, true>), dim3(grid), dim3(512), s.lds_bytes, st, a); break;
    default: hipLaunchKernelGGL((conv_wgrad_strip_kernel<3, true>), dim3(grid), dim3(512), s.lds_bytes, st, a); break;
  }
  int rc = ZS3_LAUNCH_CHECK();
  if (rc) return rc;
  if (s.splitk > 1) rc = reduce_slabs(workspace, dw, a.slab, s.splitk, st);
  return rc;
}
