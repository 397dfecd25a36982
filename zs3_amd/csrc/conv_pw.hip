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
//     steps ahead, fp32 -> bf16 hi/lo split ONCE per element (bf16-STORED activations, `A16`: 16 bytes = 8 channels per lane
//     copied as they are), ds_write into a two-stage LDS ring ([row][16 channels: hi 32 B | lo 32 B], the swizzled 64-byte
//     rows of conv_halo.hip; weights arrive pre-split from zs3_prep_weight).  They run ahead across tile boundaries: while
//     the MFMA waves finish tile j, the first K steps of tile j+1 are already in LDS / in flight.
//   * Four consumer waves (2x2, (BM/2) x 64 each) only issue ds_read_b128 + MFMA, one workgroup barrier per K step.
//   * STORE-ONLY epilogues (raw output + BatchNorm partial sums, affine, activation: every forward layer) leave the
//     accumulator registers directly: in the 32x32 MFMA layout a lane holds one column of 16 rows, so a store instruction
//     writes two full 128-byte row segments (fp32) and the per-column scale / shift are per-lane scalars; bf16 outputs pair
//     neighbouring columns through one DPP exchange per two values and store 4 bytes per lane.  Nothing is waited for -- the
//     stores drain under the next tile's MFMAs.
//   * Epilogues that LOAD per element (residual, accumulate, the fused BatchNorm-backward sums: every data-gradient launch of the
//     residual network) would stall the CU's only MFMA waves on memory latency and stay on conv_igemm.hip's kernels.  (Round 4 ran
//     them in the producer waves: correct and slower -- tools/probe/pw_lepi/ has the patch and its numbers.)
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
// -DZS3_CONV_TIMING (tools/probe/build_variant.sh + tools/probe/pw_timing.py): per-wave s_memtime split of one workgroup's life,
// written to the buffer registered with zs3_conv_pw_timing (block `buf[0]` reports)
#ifdef ZS3_CONV_TIMING
long* g_pw_timing = nullptr;
#define HT_DECL long ht_a = 0, ht_b = 0, ht_c = 0, ht_d = 0, ht_last = __builtin_readcyclecounter(); const long ht_start = ht_last;
#define HT(v) { const long t_ = __builtin_readcyclecounter(); v += t_ - ht_last; ht_last = t_; }
#define HT_STORE(w) if (tbuf && blockIdx.x == (int)tbuf[0] && lane == 0) { long* o_ = tbuf + 8 + (w) * 5; o_[0] = ht_a; o_[1] = ht_b; o_[2] = ht_c; o_[3] = ht_d; o_[4] = __builtin_readcyclecounter() - ht_start; } \
  if (tbuf && (w) == 0 && lane == 0) { long* o_ = tbuf + 64 + 4 * (long)blockIdx.x; o_[0] = ht_start; o_[1] = __builtin_readcyclecounter(); o_[2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); o_[3] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); }
#define HT_ARG , long* tbuf
#else
#define HT_DECL
#define HT(v)
#define HT_STORE(w)
#define HT_ARG
#endif

// bf16 outputs from the accumulator registers.  Lane (lr, hh) holds column lr of rows R(r) = 32 i + (r & 3) + 8 (r >> 2) + 4 hh.
// Lanes lr and lr ^ 1 swap one value per row pair (r, r + 1) through a DPP quad permutation, after which the even lane owns
// columns (lr, lr + 1) of row R(r) and the odd lane columns (lr - 1, lr) of row R(r + 1): one packed 4-byte store per two
// elements, a store instruction covers four 64-byte row segments.
template <int TM, int TN, int BM, int BN>
__device__ __forceinline__ void store_acc_direct16(const ConvArgs& p, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                                   int lane) {
  const int hh = lane >> 5, lr = lane & 31;
  const bool odd = lr & 1;
  const bool affine = (p.scale != nullptr) || (p.shift != nullptr);
  int opaque = 0;
  asm volatile("" : "+v"(opaque));   // (store_acc_direct: keeps the address arithmetic behind the K loop)
  const int lrow = wm * (BM / 2) + 4 * hh + opaque;
  bf16_t* const ybase = reinterpret_cast<bf16_t*>(p.y) + (size_t)m0 * p.ldy + n0;
  const bool full = m0 + BM <= p.M;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int colp = wn * 64 + 32 * j + (lr & ~1);          // first column of this lane's pair
    const int col = n0 + wn * 64 + 32 * j + lr;
    const int cc = col < p.ncols ? col : 0;
    const float sc = p.scale ? p.scale[cc] : 1.f, sh = p.shift ? p.shift[cc] : 0.f;
    const bool cok = n0 + colp + 1 < p.ncols;                // ncols is even (pw_ok): a pair is inside or outside
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float v0 = acc[i][j][r], v1 = acc[i][j][r + 1];
        if (affine) {
          v0 = fmaf(v0, sc, sh);
          v1 = fmaf(v1, sc, sh);
        }
        if (p.act == 1) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
        } else if (p.act == 2) {
          v0 = v0 > 0.f ? v0 : v0 * p.leak;
          v1 = v1 > 0.f ? v1 : v1 * p.leak;
        }
        const float send = odd ? v0 : v1;
        const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        const unsigned pk = cvt_pk_bf16(odd ? recv : v0, odd ? v1 : recv);
        const int rr = lrow + 32 * i + (r & 3) + 8 * (r >> 2) + (odd ? 1 : 0);
        if (cok && (full || m0 + rr < p.M)) *reinterpret_cast<unsigned*>(ybase + (size_t)rr * p.ldy + colp) = pk;
      }
  }
}

// INAFF: the producers apply x' = max(x * in_scale[c] + in_shift[c], 0) before the split (conv_common.h: ConvArgs::in_scale).
// A16  : x is stored as bf16 (plain-bf16 products): 16 bytes = 8 channels per lane and row, copied into the LDS rows.
// RAGGED: cin_valid is not a multiple of the 32-channel K step (48, 24 channels: two data-gradient launches of the decoder): lanes whose
// channels lie past cin_valid in a tile's last step read the zero page there -- two selects per load that every other launch is spared.
template <int PREC, int BM, bool INAFF = false, bool A16 = false, bool RAGGED = false>
__global__ __launch_bounds__(512, BM == 128 ? 4 : 2) void conv_pw_kernel(const ConvArgs p, const int ntiles, const int ntn HT_ARG) {
  static_assert(!(RAGGED && INAFF), "the producer-side input transform is instantiated for whole K steps only");
  static_assert(!A16 || (PREC == 1 && !INAFF), "bf16-stored input: plain bf16 products, no producer-side transform");
  constexpr int BN = PW_BN, TM = BM / 64, TN = 2;
  constexpr int ASUB = BM * 64;                     // one 16-channel sub-chunk of the activation rows
  constexpr int STAGE = 2 * ASUB + PW_BSTAGE;
  constexpr int OFF_CT = 2 * STAGE;
  constexpr int NL = PREC >= 3 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  // G <= ntiles: every workgroup owns at least one tile.  (Giving the two co-resident workgroups of a CU -- b and b + G / 2, observed with
  // tools/probe/pw_place.py -- neighbouring column tiles of the same activation rows, hoping for vector-L1 hits on the second request,
  // changed nothing: 45.28 against 45.26 ms per step, tools/probe/r5i.sh.)
  const int first = xcd_remap(blockIdx.x, G);
  const int nmine = (ntiles - first + G - 1) / G;
  const int NK = p.cin_pad >> 5;
  const int S = nmine * NK;                         // K steps of this workgroup
  const int S2 = (S + 1) & ~1;                      // the producers' schedule is unrolled by two (register sets = LDS stages)
  const int E = p.stat_partial ? 1 : 0;  // workgroup barriers of one (consumer-side) epilogue

  if (wave >= 4) {
    // ------------------------------------------------------------------ producers (256 lanes)
    // activations: fp32: lane -> (row prow + 32 r, channels 4 c8 .. +3 of the K step): 8 lanes read one row's 128 contiguous bytes
    //              bf16: lane -> (row prow16 + 64 r, channels 8 g .. +7): 4 lanes read one row's 64 contiguous bytes
    // weights    : lane -> (columns wrow, wrow + 64; 16-byte piece wq of both sub-chunks)
    constexpr int NRA = A16 ? BM / 64 : BM / 32;
    const int pl = tid - 256;
    const int prow = A16 ? pl >> 2 : pl >> 3;
    const int c8 = pl & 7, sub = A16 ? (pl >> 1) & 1 : c8 >> 2, cq = c8 & 3, half16 = pl & 1;
    const int chan0 = A16 ? (pl & 3) * 8 : c8 * 4;       // this lane's first channel inside the 32-channel K step
    constexpr int RSTEP = A16 ? 64 : 32;                  // rows between a lane's consecutive rows
    const int wrow = pl >> 2, wq = pl & 3;
    const int qoff = (wq & 1) * 16 + (wq >> 1) * 64;
    const unsigned wdst = (unsigned)(wrow * 64 + ((wq ^ ((wrow >> 2) & 3)) << 4));
    using XT = std::conditional_t<A16, bf16_t, float>;
    const XT* const xzero = reinterpret_cast<const XT*>(p.zero);
    const long rstep = RSTEP * (long)p.ldx;
    // load stream state: tile / K step of the next request.  Every load address is a per-lane POINTER that advances by a per-lane
    // increment each K step (a masked row points at the zero page and does not move): the round-3 form recomputed base + step *
    // stride with two selects per load -- 70 of the 118 VALU instructions of a producer interval, and the timing build
    // (tools/probe/pw_timing.py) showed the producers' issue time, not memory latency, bounding the K loop (1340 cycles per
    // step against 990 for the MFMA waves).
    int lt = first, lk = 0;
    const XT* ap[NRA];
    int ainc[NRA];
    const unsigned char* wp_[2];
    int winc[2];
    const float* scp = nullptr;
    // K steps for which this lane's channels lie inside cin_valid (all NK of them unless the last chunk is ragged: 48, 304 channels)
    const int klim = p.cin_valid > chan0 ? (p.cin_valid - chan0 + 31) >> 5 : 0;
    auto setup_tile = [&](int tile) {
      tile = tile < ntiles ? tile : first + (nmine - 1) * G;   // past the end: re-request the last tile (never multiplied)
      const int mt = tile / ntn, nt = tile - mt * ntn;
      const int m0 = mt * BM, n0 = nt * BN;
      const XT* abase = reinterpret_cast<const XT*>(p.x) + (long)(m0 + prow) * p.ldx + chan0;
#pragma unroll
      for (int r = 0; r < NRA; ++r) {
        const bool ok = m0 + prow + RSTEP * r < p.M;
        ap[r] = ok ? abase + r * rstep : xzero;
        ainc[r] = ok ? 32 : 0;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int col = n0 + wrow + 64 * e;
        const bool ok = col < p.ncols;
        wp_[e] = ok ? reinterpret_cast<const unsigned char*>(p.w_pk) + (size_t)col * (4 * (size_t)p.ldw) + qoff
                    : reinterpret_cast<const unsigned char*>(p.zero);
        winc[e] = ok ? 128 : 0;
      }
      if constexpr (INAFF) scp = p.in_scale + c8 * 4;
    };
    struct StepRegs {
      f32x4 a[NRA];      // four fp32 channels, or (A16) eight bf16 channels as raw bits
      u32x4 w[2][2];
      f32x4 sc, sh;      // (INAFF) scale / shift of this lane's four channels in this K step
    };
    StepRegs buf[2];
    auto load_step = [&](StepRegs& d) {
      const bool cok = !RAGGED || lk < klim;
      if (!(ZS3_PW_ABLATE & 4)) {
#pragma unroll
        for (int r = 0; r < NRA; ++r) {
          d.a[r] = *reinterpret_cast<const f32x4*>(cok ? ap[r] : xzero);
          ap[r] += ainc[r];
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
          for (int s = 0; s < 2; ++s) d.w[e][s] = *reinterpret_cast<const u32x4*>(wp_[e] + s * 32);
          wp_[e] += winc[e];
        }
        if constexpr (INAFF) {
          d.sc = *reinterpret_cast<const f32x4*>(cok ? scp : p.zero);
          d.sh = *reinterpret_cast<const f32x4*>(cok ? scp + (p.in_shift - p.in_scale) : p.zero);
          scp += 32;
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
        const int row = prow + RSTEP * r, sw = (row >> 2) & 3;
        if constexpr (A16) {   // eight consecutive channels = K half `half16` of the 16-channel sub-chunk: one 16-byte slot
          *reinterpret_cast<f32x4*>(sa + row * 64 + ((half16 ^ sw) << 4)) = s.a[r];
          continue;
        }
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
        if (PREC >= 3) *reinterpret_cast<u32x2*>(sa + row * 64 + (o ^ 32)) = lo;
      }
      unsigned char* sb = dsm + stage * STAGE + 2 * ASUB;
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c) *reinterpret_cast<u32x4*>(sb + c * 8192 + e * 4096 + wdst) = s.w[e][c];
    };

    // prologue: step 0 in stage 0; steps 1, 2 requested into register sets 1, 0
    setup_tile(first);
    load_step(buf[0]);
    write_step(buf[0], 0);
    load_step(buf[1]);
    load_step(buf[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // B_0
    HT_DECL   // producers: a = waiting for the loads of the step to be written, b = split + LDS writes + next requests, c = barriers, d = prologue
    HT(ht_d)
    // interval i (the consumers multiply step i): write step i + 1 -- register set and LDS stage (i + 1) & 1 -- then request step
    // i + 3 into the set that just emptied.  An interval that starts a tile (i = NK, 2 NK, ...) also takes part in the E barriers of
    // the previous tile's epilogue.  Two sets instead of round 3's three: the kernel now fits 128 registers and TWO workgroups
    // share a CU -- twice the requests in flight per CU, and one workgroup's epilogue / barrier waits run under the other's MFMAs.
    int nexttile = NK;
    for (int i0 = 0; i0 < S2; i0 += 2) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = i0 + r;
#ifdef ZS3_CONV_TIMING
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRA + 4 + (INAFF ? 2 : 0)) : "memory");
        HT(ht_a)
#endif
        write_step(buf[(r + 1) & 1], (r + 1) & 1);
        load_step(buf[(r + 1) & 1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HT(ht_b)
        if (i == nexttile) {
          nexttile += NK;
          if (i < S)
            for (int e = 0; e < E; ++e) __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();   // B_{i+1}: step i + 1 is in LDS
        HT(ht_c)
      }
    }
    for (int e = 0; e < E; ++e) __builtin_amdgcn_s_barrier();   // the last tile's epilogue
    HT_STORE(wave)
  } else {
    // ------------------------------------------------------------------ consumers: ds_read_b128 + MFMA, then the epilogue
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int lr = lane & 31, kh = lane >> 5;
    const unsigned swz = (unsigned)((kh ^ ((lr >> 2) & 3)) << 4);
    const unsigned aoff = (unsigned)((wm * (BM / 2) + lr) * 64) + swz;            // row block i: + 2048 i; sub-chunk: + ASUB
    const unsigned boff = (unsigned)(2 * ASUB + (wn * 64 + lr) * 64) + swz;       // column block j: + 2048 j; sub-chunk: + 8192
    HT_DECL   // consumers: a = MFMA sub-steps, b = epilogues, c = K-step barriers, d = before the first step
    f32x16 acc[TM][TN];
    // ONE set of fragments: hi / lo of TM row blocks and TN column blocks.  A 16-channel sub-step u multiplies, for the split
    // arithmetics, in three groups of TM x TN MFMAs -- G1: a_lo . b_hi, G2: a_hi . b_hi, G3: a_hi . b_lo -- so a_lo is dead after G1,
    // b_hi after G2, a_hi and b_lo after G3, and sub-step u + 1 needs them in exactly that order: each fragment of u + 1 is fetched
    // as soon as its register is free and has at least one group (TM x TN x 32 matrix-pipe cycles) to arrive.  The double-buffered
    // fragment sets of round 3 cost 64 registers; this costs none and keeps the LDS latency under the MFMAs all the same.  (Plain
    // bf16 has one group per sub-step: its fragments are re-fetched behind it and the other workgroup on the CU covers the wait.)
    bf16x8 fa[TM][2], fb[TN][2];         // [block][hi, lo]
    auto rd_a = [&](int pl, const unsigned char* st, int sc) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i][pl] = *reinterpret_cast<const bf16x8*>(st + ((aoff + i * 2048 + sc * ASUB) ^ (32u * pl)));
    };
    auto rd_b = [&](int pl, const unsigned char* st, int sc) {
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j][pl] = *reinterpret_cast<const bf16x8*>(st + ((boff + j * 2048 + sc * 8192) ^ (32u * pl)));
    };
    auto group = [&](int pa, int pb) {    // TM x TN MFMAs: a[.][pa] . b[.][pb]
      if (!(ZS3_PW_ABLATE & 2)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma16<PREC>(fa[i][pa], fb[j][pb], acc[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // sub-step on the fragments in registers; `nst` / `nsc`: LDS stage and sub-chunk of the NEXT sub-step's fragments.  `sync`: the
    // next sub-step lives in the other stage -- wait for this wave's outstanding reads of the current one, then the step barrier
    // (the producers refill the current stage behind it), before the first fetch
    // `fetch` = false: the tile's last sub-step -- nothing is fetched across the epilogue (32 registers the 128-register build
    // needs there); the next tile's first fragments are read when it starts, one exposed LDS round trip per tile
    auto substep = [&](const unsigned char* nst, int nsc, bool sync, bool fetch) {
      auto step_barrier = [&]() {
        if (sync) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          HT(ht_a)
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          HT(ht_c)
        }
      };
      if constexpr (PREC >= 3) {
        group(1, 0);
        step_barrier();
        if (fetch) rd_a(1, nst, nsc);
        __builtin_amdgcn_sched_barrier(0);
        group(0, 0);
        if (fetch) rd_b(0, nst, nsc);
        __builtin_amdgcn_sched_barrier(0);
        group(0, 1);
        if (fetch) {
          rd_a(0, nst, nsc);
          rd_b(1, nst, nsc);
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        group(0, 0);
        step_barrier();
        if (fetch) {
          rd_a(0, nst, nsc);
          rd_b(0, nst, nsc);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    __builtin_amdgcn_s_barrier();   // B_0
    asm volatile("" ::: "memory");
    float* const ctile = reinterpret_cast<float*>(dsm + OFF_CT);
    auto lds_barrier = [&]() {   // the epilogue's hazards are on the staging area only: do not drain the global stores
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    int gs = 0;   // global K-step counter (stage = gs & 1)
    HT(ht_d)
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
      {   // this tile's first fragments (its first step is in LDS since the barrier that ended the previous tile's K loop)
        const unsigned char* cur = dsm + (gs & 1) * STAGE;
#pragma unroll
        for (int pl = 0; pl < NL; ++pl) {
          rd_a(pl, cur, 0);
          rd_b(pl, cur, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      for (int k = 0; k < NK; ++k, ++gs) {
        const unsigned char* cur = dsm + (gs & 1) * STAGE;
        const unsigned char* nxt = dsm + ((gs + 1) & 1) * STAGE;
        substep(cur, 1, false, true);          // channels 0..15 of the step; fetches 16..31 of the same stage
        substep(nxt, 0, true, k + 1 < NK);     // channels 16..31; B_{gs+1} (the other stage holds step gs + 1), then fetches its 0..15
        HT(ht_a)
      }
      // ---- epilogue of this tile (consumer waves only; the producers are already filling the next tile's stages)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) unscale_acc<PREC>(acc[i][j]);   // (f16x3: the forward plane carries 2^6 w)
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
        int t2 = tid;
        asm volatile("" : "+v"(t2));   // (keeps red + tid out of the registers that live across the K loop: it was the build's one spill)
        if (t2 < BN) {
          const int col = n0 + t2;
          if (col < p.ncols) {
            p.stat_partial[((size_t)mt * 2 + 0) * p.ncols + col] = red[t2] + red[2 * BN + t2];
            p.stat_partial[((size_t)mt * 2 + 1) * p.ncols + col] = red[BN + t2] + red[3 * BN + t2];
          }
        }
      }
      // (the element type of y is the input's: one epilogue per instantiation -- with both compiled in, the unused one's register
      // pressure pushed the 128-register build into scratch; launch_pw routes mixed-type launches to the other kernels)
      if constexpr (A16) store_acc_direct16<TM, TN, BM, BN>(p, acc, m0, n0, wm, wn, lane);
      else store_acc_direct<TM, TN, BM, BN>(p, acc, m0, n0, wm, wn, lane);
      HT(ht_b)
    }
    for (int e = S; e < S2; ++e) __builtin_amdgcn_s_barrier();   // the producers' schedule is padded to a multiple of two
    HT_STORE(wave)
  }
}

// what the kernel can run at all (geometry); the epilogue / element-type questions are launch_pw's
bool pw_geom_ok(const ConvArgs& a, int bm) {
  if (a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad_h != 0 || a.pad_w != 0 || a.H != a.Ho || a.W != a.Wo) return false;
  if ((a.ldx & 3) || (a.cin_valid & 3) || (a.cin_pad & 31) || a.cin_pad < 32 || a.M <= 0) return false;
  return bm == 256 || bm == 128;
}
template <int PREC, int BM, bool INAFF = false, bool A16 = false, bool RAGGED = false>
int launch_pw_r(const ConvArgs& a, hipStream_t st) {
  static bool configured = false;
  constexpr int LDS = 2 * (2 * BM * 64 + PW_BSTAGE) + PW_CTILE;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_kernel<PREC, BM, INAFF, A16, RAGGED>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return -4;
    configured = true;
  }
  const int ntn = (a.ncols + PW_BN - 1) / PW_BN;
  const int ntiles = ((a.M + BM - 1) / BM) * ntn;
  const int wgs = BM == 128 ? 2 * g_pw_wgs : g_pw_wgs;   // 128-row tiles: 128 registers, 66 KB of LDS -> two workgroups per CU
  const int grid = ntiles < wgs ? ntiles : wgs;
#ifdef ZS3_CONV_TIMING
  hipLaunchKernelGGL((conv_pw_kernel<PREC, BM, INAFF, A16, RAGGED>), dim3(grid), dim3(512), LDS, st, a, ntiles, ntn, g_pw_timing);
#else
  hipLaunchKernelGGL((conv_pw_kernel<PREC, BM, INAFF, A16, RAGGED>), dim3(grid), dim3(512), LDS, st, a, ntiles, ntn);
#endif
  return ZS3_LAUNCH_CHECK();
}

template <int PREC, int BM, bool INAFF = false, bool A16 = false>
int launch_pw_t(const ConvArgs& a, hipStream_t st) {
  if (a.cin_valid & 31) {
    if constexpr (INAFF) return -7;
    else return launch_pw_r<PREC, BM, false, A16, true>(a, st);
  }
  return launch_pw_r<PREC, BM, INAFF, A16, false>(a, st);
}

}  // namespace

int zs3conv::pw_eligible(const ConvArgs& a, int bm) {
  if (!pw_geom_ok(a, bm)) return 0;
  return direct_epilogue(a) ? 1 : 0;
}

int zs3conv::launch_pw(const ConvArgs& a, int bm, int prec, hipStream_t st) {
  if (!pw_geom_ok(a, bm)) return -7;
  if (a.x_bf16 && (prec != 1 || (a.ldx & 7) || (a.cin_valid & 7) || a.in_scale)) return -7;
  if (a.y_bf16 && (a.ncols & 1)) return -7;
  if (a.y_bf16 != a.x_bf16) return -7;   // mixed element types (the seams of the 2-byte mode): the register-staged kernels
  if (!direct_epilogue(a)) return -7;   // residual / accumulate / fused BatchNorm-backward sums: conv_igemm.hip's kernels
  if (a.x_bf16) return bm == 256 ? launch_pw_t<1, 256, false, true>(a, st) : launch_pw_t<1, 128, false, true>(a, st);
  if (a.in_scale) {   // input transform (the producing layer's BatchNorm-apply + ReLU) in the producer waves
    if (!a.in_shift) return -1;
    if (bm == 256)
      return prec == 1 ? launch_pw_t<1, 256, true>(a, st) : prec == 4 ? launch_pw_t<4, 256, true>(a, st) : launch_pw_t<3, 256, true>(a, st);
    return prec == 1 ? launch_pw_t<1, 128, true>(a, st) : prec == 4 ? launch_pw_t<4, 128, true>(a, st) : launch_pw_t<3, 128, true>(a, st);
  }
  if (bm == 256) return prec == 1 ? launch_pw_t<1, 256>(a, st) : prec == 4 ? launch_pw_t<4, 256>(a, st) : launch_pw_t<3, 256>(a, st);
  return prec == 1 ? launch_pw_t<1, 128>(a, st) : prec == 4 ? launch_pw_t<4, 128>(a, st) : launch_pw_t<3, 128>(a, st);
}

// Whether tile_cfg 51 / 52 can run this convolution's GEOMETRY (1x1, stride 1, no padding; callers fall back to tile_cfg 31
// otherwise).  Store-only epilogues only (zs3_conv_igemm returns -7 otherwise).
extern "C" int zs3_conv_pw_ok(int N, int H, int W, int Ho, int Wo, int cin_pad, int cin_valid, int ldx, int KH, int KW, int stride,
                              int pad_h, int pad_w, int tile_cfg) {
  ConvArgs a{};
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.cin_pad = cin_pad; a.cin_valid = cin_valid; a.ldx = ldx;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  a.M = N * Ho * Wo;
  return pw_geom_ok(a, tile_cfg == 52 ? 128 : 256) ? 1 : 0;
}

#ifdef ZS3_CONV_TIMING
extern "C" int zs3_conv_pw_timing(long* buf) {   // device int64 buffer: [0] = reporting block, [8 + 5 wave ..] = that block's per-wave cycle split
  g_pw_timing = buf;
  return 0;
}
#endif

// Persistent workgroups per launch of tile_cfg 51 / 52 (default 256 = one per CU); returns the previous value.  Tests set a
// small number so that small problems exercise the cross-tile pipeline.
extern "C" int zs3_conv_pw_set_wgs(int wgs) {
  const int old = g_pw_wgs;
  if (wgs > 0) g_pw_wgs = wgs;
  return old;
}
