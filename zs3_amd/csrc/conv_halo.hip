// Strip-resident multi-tap convolution for gfx950 (MI355X): forward and data-gradient of every stride-1, same-size
// KHxKW convolution (the 3x3 / dilated 3x3 layers of the network), tile_cfg 41 (256-row tiles) and 42 (192-row tiles).
//
// The implicit-GEMM kernels of conv_igemm.hip fetch the A operand (256 output pixels x 32 channels) from L2 once per
// filter tap: nine times per channel chunk for a 3x3 layer, and the measured bound of their K loop is exactly that
// L2 -> LDS path (48 KB per K step against the ~23 B/clk/CU it delivers).  Here the nine taps share one staged copy:
//
//   * In flattened pixel space (m = (n*H + y)*W + x) tap (dy,dx) of output pixel m reads input pixel m + dy*W + dx, so a
//     tile of BM consecutive output pixels needs the contiguous *strip* of input pixels [m0 + off_min, m0 + BM + off_max).
//     For a 3x3, dilation-d layer that is BM + 2d(W+1) rows: 1.27x the tile at W = 33, 2x at W = 129 -- instead of 9x.
//   * Four producer waves load the strip of one channel chunk (16 channels in bf16x3 mode, 32 in plain-bf16 mode) from
//     L2 into registers ONCE, split it into bf16 hi/lo there and write 64-byte rows [hi | lo] into LDS (double buffered:
//     the strip of chunk c+1 is converted while the nine taps of chunk c are multiplied).  The fp32 -> bf16 split -- 65 %
//     of the issue slots of the LDS-DMA kernel's consumers -- is done once per element instead of once per tap and has
//     left the MFMA waves altogether.
//   * The same waves stream the weight tile of every K step (128 columns x 64 B, the hi/lo lines of zs3_prep_weight):
//     plain loads three tiles ahead, ds_write_b128 into a two-slot ring one step before the tile is multiplied.
//   * Four consumer waves (2x2, (BM/2)x64 wave tiles) do nothing but ds_read_b128 + MFMA: a tap is a *shifted window*
//     of the strip (row + dy*W + dx), image-border taps are redirected per lane to a zero row (9-bit mask per row,
//     computed once per tile).  Rows are 64 B apart; 16-byte chunk q of strip / weight row r sits at q ^ ((r>>2)&3), so
//     the 16 lanes of a ds_read_b128 group (16 consecutive rows, any window shift) cover all 64 banks.
//
// L2 -> CU bytes per K16 step: 8 KB of weights + strip/9 (2.7 KB at W = 33) against 24 KB before.
//
// The epilogue (BatchNorm partial sums, affine / residual / activation / accumulate, BN-backward sums) is the shared
// one of conv_common.h.  Replaces: the 3x3 nn.Conv2d of resnet.py:18-26 (layer 2-4 conv2), aspp.py:11-19 (atrous
// branches), decoder.py:15-24 (last_conv) and their data gradients.
#include <type_traits>

#include "conv_common.h"
#include "zs3hip.h"

namespace {

struct HaloGeom {
  int off_min;     // smallest tap offset in flattened pixels (<= 0)
  int s_pad;       // strip rows (multiple of 64)
  int npass;       // 64-row conversion passes per strip
  int npg;         // passes per K-step interval (<= 4)
  int nch;         // channel chunks
  int ns;          // K steps, rounded up to even (an odd tail step multiplies zeros)
  int T;           // filter taps
  int sgn;         // +1 forward, -1 dgrad (tap offsets mirrored)
  int toff0, tstep_col, tstep_row;   // flattened-pixel offset of tap 0; its change to the next tap in a row / to the next row's first tap
  int lds_bytes;
  int ntiles;      // row tiles x column tiles of the launch (the grid, unless zs3_conv_halo_set_wgs caps it)
};

constexpr int HALO_BSLOT = 8192, HALO_NSLOT = 2;   // weight tiles in LDS: the one being multiplied and the next (three more are in flight in registers)
constexpr int HALO_OFF_ZERO = HALO_NSLOT * HALO_BSLOT;   // 256 B of zeros: where masked taps read
constexpr int HALO_OFF_STRIP = HALO_OFF_ZERO + 256;
constexpr int HALO_MAXP = 3;

// -DZS3_HALO_ABLATE=n builds (tools/probe/build_variant.sh; timing probes, wrong results): 1 = no strip refills, 2 = no weight
// refills -- compile-time switches: a run-time test inside the unrolled producer schedule makes hipcc wait vmcnt(0).
#ifndef ZS3_HALO_ABLATE
#define ZS3_HALO_ABLATE 0
#endif
// -DZS3_CONV_TIMING (tools/probe/build_variant.sh + tools/probe/halo_timing.py): per-wave s_memtime split of block 0's K loop,
// written to the buffer passed as `res` when act == 99
#ifdef ZS3_CONV_TIMING
#define HT_DECL long ht_a = 0, ht_b = 0, ht_c = 0, ht_last = __builtin_readcyclecounter();
#define HT(v) { const long t_ = __builtin_readcyclecounter(); v += t_ - ht_last; ht_last = t_; }
#define HT_STORE(w) if (p.act == 99 && blockIdx.x == 0 && lane == 0) { long* o_ = reinterpret_cast<long*>(const_cast<float*>(p.res)) + (w) * 3; o_[0] = ht_a; o_[1] = ht_b; o_[2] = ht_c; }
#else
#define HT_DECL
#define HT(v)
#define HT_STORE(w)
#endif

// A16: the input tensor is stored as bf16 ([pixel][channel], ldx in elements) -- plain-bf16 arithmetic only (PREC = 1): the producers
// copy 16 bytes per lane and row from L2 into the strip, no conversion, half the bytes.
// INAFF: the producers apply x' = max(x * in_scale[c] + in_shift[c], 0) before the split -- the BatchNorm-apply + ReLU of the layer
// that produced x, whose activation tensor then never exists in memory (two VALU operations per element in waves that have
// the slack; an instantiation of its own, so the plain kernel's code is untouched).
template <int PREC, int BM, int NPG, bool A16 = false, bool INAFF = false, bool PERSIST = false>
__global__ __launch_bounds__(512) void conv_halo_kernel(const ConvArgs p, const HaloGeom g) {
  static_assert(!A16 || PREC == 1, "bf16-stored input: plain bf16 products only");
  static_assert(!(A16 && INAFF), "the input transform reads fp32 storage");
  constexpr int BN = 128, TM = BM / 64, TN = 2, CH = PREC >= 3 ? 16 : 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = (p.ncols + BN - 1) / BN;
  // One tile per workgroup (grid = tiles), or -- PERSIST, zs3_conv_halo_set_wgs -- a grid of at most that many workgroups that walk the
  // tiles t, t + grid, ...: every workgroup of the launch is handed out at once.  An instantiation of its own: the loop around the tile
  // costs the allocator its last registers (230 -> 256 with 32-124 spilled), the one-tile kernel stays as it was.  A kernel of another hardware queue starts beside such a
  // launch almost as if the chip were idle, beside a many-round launch it waits ~35 us (tools/probe/queue_gate.py: a chain of 5 us
  // launches runs at 5.9 us per launch beside the 182-tile layer-3 launch, at 38 us beside the 2774-tile decoder launch); the GMMN
  // step's frozen feature pass runs beside the generator's update chain (gmmn_trainer.FEATURE_HALO_WGS).
  int t = blockIdx.x;
  do {
  const int tile = PERSIST ? xcd_remap(t, g.ntiles) : xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / ntn, nt = tile - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int T = g.T, NS = g.nch * T, NSR = g.ns;
  const int strip_bytes = g.s_pad * 64;
  const bool producer = wave >= 4;
  const int wm = (wave >> 1) & 1, wn = wave & 1;

  f32x16 acc[TM][TN];

  if (wave >= 4) {
    // ------------------------------------------------------------------ producers: four identical waves (256 lanes)
    // Everything is a plain global_load into registers followed, a few K steps later, by a ds_write: loads of one wave
    // pipeline freely and hipcc counts their vmcnt waits itself (global_load_lds, tried first for the weight tiles, costs
    // the issuing wave ~190 cycles per 1-KB piece -- every piece re-programs M0 -- and two waves x 4 pieces per K step were
    // the last to arrive at every barrier).  The nine intervals of a channel chunk are unrolled, so every register set
    // is indexed statically.
    //   weights: tile u = (chunk, tap) is 128 columns x 64 B; a lane owns two 16-byte chunks of it.  Loaded in interval
    //            u-4 (three tiles in flight), written to LDS slot u&1 in interval u-1, read by the consumers after
    //            barrier B_u.
    //   strip  : 64-row passes (4 lanes per row, each 4 (bf16x3) or 8 (plain bf16) consecutive channels); the strip of
    //            chunk c+1 is fetched in six groups of NPG passes during the intervals of chunk c: group k loaded in
    //            interval k, split to bf16 hi/lo and written in interval k+3.  A group that reaches past the strip's last
    //            pass repeats that pass (same data, same rows).
    constexpr int NV = PREC >= 3 ? 1 : 2;
    constexpr int DIST = 3, NGRP = 9 - DIST;
    const int pl = tid - 256, prow = pl >> 2, cq = pl & 3;
    const long Mtot = (long)p.N * p.H * p.W;
    const long q0 = (long)m0 + g.off_min;
    f32x4 sbuf[DIST][NPG][NV];
    u32x4 wbuf[3][2];
    auto load_pass = [&](f32x4 (&dstv)[NV], int pass, int c) {
      pass = pass < g.npass ? pass : g.npass - 1;
      long q = q0 + pass * 64 + prow;
      q = q < 0 ? 0 : (q >= Mtot ? Mtot - 1 : q);
      const int ch0 = c * CH + cq * (CH / 4);
      if constexpr (A16) {   // 8 bf16 channels = 16 bytes per lane
        const unsigned short* src = reinterpret_cast<const unsigned short*>(p.x) + q * p.ldx + ch0;
        const void* s = ch0 < p.cin_valid ? static_cast<const void*>(src) : static_cast<const void*>(p.zero);
        dstv[0] = *reinterpret_cast<const f32x4*>(s);
      } else {
        const float* src = p.x + q * p.ldx + ch0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const float* s = ch0 + 4 * v < p.cin_valid ? src + 4 * v : p.zero;
          dstv[v] = *reinterpret_cast<const f32x4*>(s);
        }
      }
    };
    // (INAFF) scale / shift of this lane's channels in the chunk whose strip is being written; channels past cin_valid get 0 / 0
    f32x4 aff_sc[NV], aff_sh[NV];
    auto load_aff = [&](int c) {
      if constexpr (INAFF) {
        const int ch0 = c * CH + cq * (CH / 4);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const bool ok = ch0 + 4 * v < p.cin_valid;
          aff_sc[v] = *reinterpret_cast<const f32x4*>(ok ? p.in_scale + ch0 + 4 * v : p.zero);
          aff_sh[v] = *reinterpret_cast<const f32x4*>(ok ? p.in_shift + ch0 + 4 * v : p.zero);
        }
      }
    };
    auto write_pass = [&](const f32x4 (&srcv0)[NV], int pass, int sb) {
      pass = pass < g.npass ? pass : g.npass - 1;
      const int s = pass * 64 + prow, sw = (s >> 2) & 3;
      unsigned char* row = dsm + HALO_OFF_STRIP + sb * strip_bytes + s * 64;
      f32x4 srcv[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        srcv[v] = srcv0[v];
        if constexpr (INAFF) srcv[v] = affine_relu4(srcv[v], aff_sc[v], aff_sh[v]);
      }
      if (PREC >= 3) {
        u32x2 hi, lo;
        unsigned h, l;
        split_pair<PREC>(srcv[0][0], srcv[0][1], h, l); hi[0] = h; lo[0] = l;
        split_pair<PREC>(srcv[0][2], srcv[0][3], h, l); hi[1] = h; lo[1] = l;
        const int o = (((cq >> 1) ^ sw) << 4) + (cq & 1) * 8;
        *reinterpret_cast<u32x2*>(row + o) = hi;
        *reinterpret_cast<u32x2*>(row + (o ^ 32)) = lo;
      } else if constexpr (A16) {
        u32x4 hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) hi[e] = __float_as_uint(srcv[0][e]);
        *reinterpret_cast<u32x4*>(row + ((cq ^ sw) << 4)) = hi;
      } else {
        u32x4 hi;
        hi[0] = cvt_pk_bf16(srcv[0][0], srcv[0][1]);
        hi[1] = cvt_pk_bf16(srcv[0][2], srcv[0][3]);
        hi[2] = cvt_pk_bf16(srcv[NV - 1][0], srcv[NV - 1][1]);
        hi[3] = cvt_pk_bf16(srcv[NV - 1][2], srcv[NV - 1][3]);
        *reinterpret_cast<u32x4*>(row + ((cq ^ sw) << 4)) = hi;
      }
    };
    // weight tile: lane -> (tile row wr + 64 e, chunk cq), e = 0, 1; masked columns read the zero page
    const int qoff = PREC >= 3 ? (cq & 1) * 16 + (cq >> 1) * 64 : cq * 16;
    const unsigned char* wptr[2];
    int wstep[2];
    unsigned wdst[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int wr = prow + 64 * e, col = n0 + wr;
      const bool ok = col < p.ncols;
      wptr[e] = ok ? reinterpret_cast<const unsigned char*>(p.w_pk) + (size_t)col * (4 * (size_t)p.ldw) + qoff
                   : reinterpret_cast<const unsigned char*>(p.zero);
      wstep[e] = ok ? 1 : 0;
      wdst[e] = (unsigned)(wr * 64 + ((cq ^ ((wr >> 2) & 3)) << 4));
    }
    auto load_w = [&](u32x4 (&dstv)[2], int c, int t) {   // tile (c, t); t may run past 8 into the next chunk
      if (t >= 9) {
        t -= 9;
        ++c;
      }
      const int kofs = t * p.cin_pad + c * CH;
      const int uoff = (kofs >> 5) * 128 + (PREC >= 3 ? ((kofs >> 4) & 1) * 32 : 0);
#pragma unroll
      for (int e = 0; e < 2; ++e) dstv[e] = *reinterpret_cast<const u32x4*>(wptr[e] + (size_t)uoff * wstep[e]);
    };
    auto write_w = [&](const u32x4 (&srcv)[2], int slot) {
#pragma unroll
      for (int e = 0; e < 2; ++e) *reinterpret_cast<u32x4*>(dsm + slot * HALO_BSLOT + wdst[e]) = srcv[e];
    };
    if (pl < 16) *reinterpret_cast<u32x4*>(dsm + HALO_OFF_ZERO + pl * 16) = u32x4{0u, 0u, 0u, 0u};
    // ---- prologue: weight tiles 0..3 requested, strip 0 and tile 0 in LDS (NS >= 9: the four tiles exist)
    load_aff(0);
    load_w(wbuf[0], 0, 0);
    load_w(wbuf[1], 0, 1);
    load_w(wbuf[2], 0, 2);
#pragma unroll
    for (int g0 = 0; g0 < NGRP; g0 += DIST) {
#pragma unroll
      for (int s = 0; s < DIST; ++s)
#pragma unroll
        for (int k = 0; k < NPG; ++k) load_pass(sbuf[s][k], (g0 + s) * NPG + k, 0);
      if (g0 == 0) {
        write_w(wbuf[0], 0);
        load_w(wbuf[0], 0, 3);
      }
#pragma unroll
      for (int s = 0; s < DIST; ++s)
#pragma unroll
        for (int k = 0; k < NPG; ++k) write_pass(sbuf[s][k], (g0 + s) * NPG + k, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // B_0
    HT_DECL
    // ---- interval s = 9c + t (between B_s and B_{s+1}): write tile s+1, request tile s+4 into the set it leaves,
    //      convert strip group t-3 of chunk c+1, request group t
    for (int c = 0; c + 1 < g.nch; ++c) {
      const int sb = (c + 1) & 1;
      load_aff(c + 1);   // every strip write of this iteration belongs to chunk c + 1
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        write_w(wbuf[(t + 1) % 3], (c + t + 1) & 1);
        if (!(ZS3_HALO_ABLATE & 2)) load_w(wbuf[(t + 1) % 3], c, t + 4);
        if (!(ZS3_HALO_ABLATE & 1)) {
          if (t >= DIST) {
#pragma unroll
            for (int k = 0; k < NPG; ++k) write_pass(sbuf[(t - DIST) % DIST][k], (t - DIST) * NPG + k, sb);
          }
          if (t < NGRP) {
#pragma unroll
            for (int k = 0; k < NPG; ++k) load_pass(sbuf[t % DIST][k], t * NPG + k, c + 1);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HT(ht_a)
        __builtin_amdgcn_s_barrier();   // B_{s+1}
        HT(ht_c)
      }
    }
    {   // last chunk: no further strip; tiles s+1 / s+4 exist while they stay inside the chunk
      const int c = g.nch - 1;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t + 1 < 9) write_w(wbuf[(t + 1) % 3], (c + t + 1) & 1);
        if (t + 4 < 9) load_w(wbuf[(t + 1) % 3], c, t + 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HT(ht_a)
        __builtin_amdgcn_s_barrier();   // B_{s+1}
        HT(ht_c)
      }
    }
    if (NSR > NS) __builtin_amdgcn_s_barrier();   // the padding step of an odd K loop (the consumers multiply the zero row)
    HT_STORE(wave)
  } else {
    // ------------------------------------------------------------------ consumers: ds_read_b128 + MFMA only
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, kh = lane >> 5;
    // tap masks of this lane's TM rows (bit t: tap t reads inside the image)
    unsigned vmask[TM];
    {
      const int hw = p.H * p.W;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 32 + lr;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / hw, rem = mm - n * hw;
        const int y = rem / p.W, x = rem - y * p.W;
        unsigned mk = 0u;
        for (int t = 0; t < T; ++t) {
          const int th = t / p.KW, tw = t - th * p.KW;
          const int iy = y + g.sgn * (th * p.dil - p.pad_h), ix = x + g.sgn * (tw * p.dil - p.pad_w);
          if (ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mk |= 1u << t;
        }
        vmask[i] = mk;
      }
    }
    const int rb = wm * (BM / 2) + lr - g.off_min;                      // strip row of row block 0 at tap offset 0
    const unsigned boff = (unsigned)((wn * 64 + lr) * 64 + ((kh ^ ((lr >> 2) & 3)) << 4));   // weight-tile row of column block 0
    const unsigned zaddr = HALO_OFF_ZERO;

    bf16x8 a_hi[TM], a_lo[TM];        // one buffer per row block: fragment f is read two sub-steps before its MFMAs
    bf16x8 b_hi[2][TN], b_lo[2][TN];
    // K-step state (wave-uniform): tap (rt, column rtw) / chunk rc / flattened-pixel offset rtoff of the step whose window
    // is computed next -- all scalar, advanced incrementally (no division, no table)
    int rt = 0, rtw = 0, rc = 0, rtoff = g.toff0;
    unsigned a0 = 0u, tbit = 0u;     // strip address of row block 0 / mask bit of the step being read
    unsigned a0n = 0u, tbitn = 0u;   // the same for the step after it (computed in sub-step 0, in the MFMA shadow)
    auto advance_read = [&]() {
      const bool wcol = rtw + 1 == p.KW, wtap = rt + 1 == T;
      rtoff = wtap ? g.toff0 : rtoff + (wcol ? g.tstep_row : g.tstep_col);
      rtw = (wcol || wtap) ? 0 : rtw + 1;
      rt = wtap ? 0 : rt + 1;
      rc += wtap ? 1 : 0;
    };
    auto next_window = [&]() {       // (rtoff, rt, rc) -> a0n, tbitn
      const int s0 = rb + rtoff;
      a0n = (unsigned)(HALO_OFF_STRIP + (rc & 1) * strip_bytes + s0 * 64 + ((kh ^ ((s0 >> 2) & 3)) << 4));
      tbitn = rc < g.nch ? 1u << rt : 0u;   // the padding step of an odd K loop reads the zero row
    };
    // A fragment f: its (masked) address is computed in one MFMA gap, its two reads are issued in two later gaps -- a gap
    // hides ~32 cycles of issue (2 LDS reads or ~6 VALU); more in one gap delays the next MFMA
    unsigned aaddr = 0u;
    auto addr_a = [&](int f, unsigned base, unsigned bit) { aaddr = (vmask[f] & bit) ? base + f * 2048 : zaddr; };
    auto read_a_hi = [&](int f) { a_hi[f] = *reinterpret_cast<const bf16x8*>(dsm + aaddr); };
    auto read_a_lo = [&](int f) { a_lo[f] = *reinterpret_cast<const bf16x8*>(dsm + (aaddr ^ 32u)); };
    auto read_b = [&](int set, int slot, int j) {
      const unsigned addr = slot * HALO_BSLOT + boff + j * 2048;
      b_hi[set][j] = *reinterpret_cast<const bf16x8*>(dsm + addr);
      b_lo[set][j] = *reinterpret_cast<const bf16x8*>(dsm + (addr ^ 32u));
    };
    int slot = 0;
    HT_DECL
    // One K step = TM sub-steps (row blocks) of 3*TN (bf16x3) or 2*TN MFMAs.  Sub-step i runs from registers; in its MFMA
    // shadow the fragment of sub-step i+2 is read: this step's row block i+2, or -- in the last two sub-steps, after the
    // step barrier -- the next step's row blocks 0 and 1 and its weight fragments (the other register set).  The barrier
    // (next weight tile landed, next strip complete) therefore sits before sub-step TM-2.  Every MFMA slot is pinned
    // (sched_barrier): hipcc otherwise hoists all LDS reads and issues the MFMAs as one clump.
    auto step = [&](auto setc) {
      constexpr int SET = decltype(setc)::value;
      constexpr int NM = (PREC >= 3 ? 3 : 2) * TN;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (i == TM - 2) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          HT(ht_a)
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          HT(ht_c)
          a0 = a0n;
          tbit = tbitn;
          slot = slot == HALO_NSLOT - 1 ? 0 : slot + 1;
        }
        const bf16x8 ah = a_hi[i], al = a_lo[i];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          const int pr = m / TN, j = m % TN;
          if (PREC >= 3)
            acc[i][j] = mfma16<PREC>(pr == 0 ? al : ah, pr == 1 ? b_lo[SET][j] : b_hi[SET][j], acc[i][j]);
          else
            acc[i][j] = mfma16<PREC>(pr == 0 ? ah : al, pr == 0 ? b_hi[SET][j] : b_lo[SET][j], acc[i][j]);
          // fillers of this gap.  Sub-step i fetches fragment (i + 2) % TM (its address was computed a sub-step ago) and
          // computes the address of fragment (i + 3) % TM; fragments fetched in sub-steps TM-2 and TM-1 belong to the next
          // step, so sub-step TM-3 computes its address from the NEXT window (a0n / tbitn, ready since sub-step 0).
          if (m == 0) read_a_hi((i + 2) % TM);
          if (m == 1) read_a_lo((i + 2) % TM);
          if (i == 0 && m == 2) advance_read();
          if (i == 0 && m == NM - 1) next_window();
          if (i == TM - 2 && m >= 2 && m < 2 + TN) read_b(SET ^ 1, slot, m - 2);
          if (m == NM - 1 && i > 0) addr_a((i + 3) % TM, i == TM - 3 ? a0n : a0, i == TM - 3 ? tbitn : tbit);
          __builtin_amdgcn_sched_barrier(0);
          if (m == NM - 1 && i == 0) {   // (after next_window(): TM = 3 needs the next window here)
            addr_a(3 % TM, TM == 3 ? a0n : a0, TM == 3 ? tbitn : tbit);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
    __builtin_amdgcn_s_barrier();   // B_0: strip 0 and weight tile 0 are in LDS
    asm volatile("" ::: "memory");
    // (s_setprio(1) for the MFMA waves: 45.84 / 45.97 ms per step against 45.74 / 45.76 without -- not kept)
    next_window();
    a0 = a0n;
    tbit = tbitn;
    addr_a(0, a0, tbit);
    read_a_hi(0);
    read_a_lo(0);
    addr_a(1, a0, tbit);
    read_a_hi(1);
    read_a_lo(1);
#pragma unroll
    for (int j = 0; j < TN; ++j) read_b(0, 0, j);
    addr_a(2 % TM, a0, tbit);   // what sub-step 0 fetches
    __builtin_amdgcn_sched_barrier(0);
    for (int s = 0; s < NSR; s += 2) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
    }
    HT_STORE(wave)
  }

  // ---------------------------------------------------------------------- epilogue (all eight waves store rows)
  __syncthreads();   // every DMA has landed and been consumed; nobody reads the operand LDS any more
  if (!producer) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) unscale_acc<PREC>(acc[i][j]);   // (f16x3: the forward plane carries 2^6 w)
  }
  float* ctile = reinterpret_cast<float*>(dsm);
  if (p.stat_partial) {
    float* red = ctile;
    if (!producer) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float s = 0.f, q2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r];
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
    }
    __syncthreads();
    if (tid < BN) {
      const int col = n0 + tid;
      if (col < p.ncols) {
        p.stat_partial[((size_t)mt * 2 + 0) * p.ncols + col] = red[tid] + red[2 * BN + tid];
        p.stat_partial[((size_t)mt * 2 + 1) * p.ncols + col] = red[BN + tid] + red[3 * BN + tid];
      }
    }
  }
  constexpr int LDC = BN + 4;
  const bool affine = (p.scale != nullptr) || (p.shift != nullptr);
  constexpr int C4 = BN / 4, RPP = 512 / C4;
  const int c4 = tid % C4, r0 = tid / C4;
  const int col = n0 + c4 * 4;
  const bool vec = ((p.ldy & 3) == 0) && ((p.ncols & 3) == 0) && (!p.res || (p.ldr & 3) == 0);
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (col + e < p.ncols) {
      if (p.scale) sc[e] = p.scale[col + e];
      if (p.shift) sh[e] = p.shift[col + e];
    }
  f32x4 bs_s = {0.f, 0.f, 0.f, 0.f}, bs_q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    if (!producer && wm == half) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            ctile[row * LDC + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
          }
    }
    __syncthreads();
    store_tile_rows<RPP, (BM == 256 ? 2 : 4)>(p, ctile, LDC, m0 + half * (BM / 2), BM / 2, col, c4, r0, sc, sh, affine, vec, bs_s, bs_q);
  }
  if (p.bs_partial) finish_bwd_stats<BN, RPP, 512>(p, ctile, tid, c4, r0, mt, n0, bs_s, bs_q);
  if constexpr (!PERSIST) break;
  t += gridDim.x;
  if (t >= g.ntiles) break;
  __syncthreads();   // (a further tile: its producers overwrite the LDS the epilogue staged through)
  } while (true);
}

int g_halo_wgs = 0;      // zs3_conv_halo_set_wgs: > 0 = launches with more tiles run on this many workgroups (a multiple of 8: xcd_remap)

bool halo_geometry(const ConvArgs& a, int bm, int prec, HaloGeom* out) {
  const int T = a.KH * a.KW;
  if (a.stride != 1 || a.H != a.Ho || a.W != a.Wo || T != 9) return false;   // the strip schedule is unrolled for 9 taps
  if ((a.ldx & 3) || (a.cin_valid & 3) || (a.cin_pad & 31) || a.M <= 0) return false;
  if (bm != 256 && bm != 192) return false;
  HaloGeom g;
  g.T = T;
  g.sgn = a.dgrad ? -1 : 1;
  long omin = 0, omax = 0;
  for (int th = 0; th < a.KH; ++th)
    for (int tw = 0; tw < a.KW; ++tw) {
      const long o = (long)g.sgn * ((long)(th * a.dil - a.pad_h) * a.W + (tw * a.dil - a.pad_w));
      omin = o < omin ? o : omin;
      omax = o > omax ? o : omax;
    }
  const long S = bm + omax - omin;
  if (S > 4096) return false;
  g.off_min = (int)omin;
  g.toff0 = g.sgn * (-a.pad_h * a.W - a.pad_w);
  g.tstep_col = g.sgn * a.dil;
  g.tstep_row = g.sgn * (a.dil * a.W - (a.KW - 1) * a.dil);
  g.s_pad = (int)((S + 63) / 64 * 64);
  g.npass = g.s_pad / 64;
  g.npg = (g.npass + 5) / 6;   // six load groups per chunk (conv_halo_kernel: NGRP)
  if (g.npg > HALO_MAXP) return false;
  const int ch = prec >= 3 ? 16 : 32;
  g.nch = (a.cin_valid + ch - 1) / ch;
  if (g.nch * ch > a.cin_pad) return false;
  g.ns = (g.nch * T + 1) & ~1;
  const int epi = (bm / 2) * (128 + 4) * 4;
  const int loop = HALO_OFF_STRIP + 2 * g.s_pad * 64;
  g.lds_bytes = loop > epi ? loop : epi;
  if (g.lds_bytes > 160 * 1024) return false;
  if (out) *out = g;
  return true;
}

template <int PREC, int BM, int NPG, bool A16 = false, bool INAFF = false>
int launch_halo_n(const ConvArgs& a, const HaloGeom& g, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<PREC, BM, NPG, A16, INAFF>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return -4;
    configured = true;
  }
  HaloGeom gg = g;
  gg.ntiles = ((a.M + BM - 1) / BM) * ((a.ncols + 127) / 128);
  // the tile-walking form exists for the 192-row forward instantiations (f16x3 and plain bf16): what a frozen feature pass launches
  if constexpr (BM == 192 && (PREC == 4 || PREC == 1)) {
    // (every launch with more tiles than the cap, also the 182-tile layer-3 launches that are handed out in one go as they are: walking
    // their tiles on 128 workgroups makes them 70 % longer, 140 against 82 us, and the GMMN step 0.6 ms shorter -- the CUs they leave are
    // what the loop, the step's critical path, runs on; capping only the many-round launches: 21.8-21.9 against 21.2-21.4 ms)
    if (g_halo_wgs > 0 && gg.ntiles > g_halo_wgs && !a.dgrad) {
      static bool configured_p = false;
      if (!configured_p) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<PREC, BM, NPG, A16, INAFF, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
          return -4;
        configured_p = true;
      }
      hipLaunchKernelGGL((conv_halo_kernel<PREC, BM, NPG, A16, INAFF, true>), dim3(g_halo_wgs), dim3(512), g.lds_bytes, st, a, gg);
      return ZS3_LAUNCH_CHECK();
    }
  }
  const int grid = gg.ntiles;
  hipLaunchKernelGGL((conv_halo_kernel<PREC, BM, NPG, A16, INAFF>), dim3(grid), dim3(512), g.lds_bytes, st, a, gg);
  return ZS3_LAUNCH_CHECK();
}
template <int PREC, int BM, bool A16 = false, bool INAFF = false>
int launch_halo_t(const ConvArgs& a, const HaloGeom& g, hipStream_t st) {
  switch (g.npg) {
    case 1: return launch_halo_n<PREC, BM, 1, A16, INAFF>(a, g, st);
    case 2: return launch_halo_n<PREC, BM, 2, A16, INAFF>(a, g, st);
    case 3: return launch_halo_n<PREC, BM, 3, A16, INAFF>(a, g, st);
  }
  return -7;
}

}  // namespace

int zs3conv::halo_eligible(const ConvArgs& a, int bm, int prec) {   // 0: not eligible, else NPG (the kernel instantiation's third template argument)
  HaloGeom g;
  return halo_geometry(a, bm, prec, &g) ? g.npg : 0;
}

int zs3conv::launch_halo(const ConvArgs& a, int bm, int prec, hipStream_t st) {
  HaloGeom g;
  if (!halo_geometry(a, bm, prec, &g)) return -7;
  if (a.x_bf16) {   // bf16-stored input (tile_cfg 141 / 142): plain-bf16 products, 8-channel granularity
    if (prec != 1 || (a.ldx & 7) || (a.cin_valid & 7) || a.in_scale) return -7;
    return bm == 256 ? launch_halo_t<1, 256, true>(a, g, st) : launch_halo_t<1, 192, true>(a, g, st);
  }
  if (a.in_scale) {   // input transform (the producing layer's BatchNorm-apply + ReLU) in the producer waves
    if (!a.in_shift) return -1;
    if (bm == 256)
      return prec == 1 ? launch_halo_t<1, 256, false, true>(a, g, st)
             : prec == 4 ? launch_halo_t<4, 256, false, true>(a, g, st) : launch_halo_t<3, 256, false, true>(a, g, st);
    return prec == 1 ? launch_halo_t<1, 192, false, true>(a, g, st)
           : prec == 4 ? launch_halo_t<4, 192, false, true>(a, g, st) : launch_halo_t<3, 192, false, true>(a, g, st);
  }
  if (bm == 256)
    return prec == 1 ? launch_halo_t<1, 256>(a, g, st) : prec == 4 ? launch_halo_t<4, 256>(a, g, st) : launch_halo_t<3, 256>(a, g, st);
  return prec == 1 ? launch_halo_t<1, 192>(a, g, st) : prec == 4 ? launch_halo_t<4, 192>(a, g, st) : launch_halo_t<3, 192>(a, g, st);
}

// Workgroups per launch of the strip-resident kernel: 0 (default) = one per tile; n > 0 = launches with more than n tiles run on n
// workgroups that walk the tiles (every workgroup handed out at once: see the kernel).  Returns the previous setting.
extern "C" int zs3_conv_halo_set_wgs(int wgs) {
  const int old = g_halo_wgs;
  g_halo_wgs = wgs > 0 ? (wgs + 7) / 8 * 8 : 0;
  return old;
}

// Whether tile_cfg 41 / 42 can run this convolution (callers fall back to tile_cfg 31 otherwise): 0 = no, otherwise the number
// of strip passes per K step the launch will use = the third template argument of the conv_halo_kernel instantiation it runs
// (what a profiler lists it under).
extern "C" int zs3_conv_halo_ok(int N, int H, int W, int Ho, int Wo, int cin_pad, int cin_valid, int ldx, int KH, int KW,
                                int stride, int pad_h, int pad_w, int dil, int dgrad, int prec, int tile_cfg) {
  ConvArgs a{};
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.cin_pad = cin_pad; a.cin_valid = cin_valid; a.ldx = ldx;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.dil = dil;
  a.dgrad = dgrad; a.M = N * Ho * Wo;
  return zs3conv::halo_eligible(a, tile_cfg == 42 ? 192 : 256, prec);
}
