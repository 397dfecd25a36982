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
//   * LOADING epilogues (round 4, `LEPI`: residual / lazily masked skip gradient, accumulate, the fused BatchNorm-backward
//     sums -- every data-gradient launch of the residual network) would stall the CU's only MFMA waves on memory latency.
//     They are the PRODUCERS' job: the consumers drop their accumulators into an LDS tile (64 ds_write_b32 per lane, no
//     wait) and go on multiplying the next tile; the producer waves -- which issue the kernel's global loads anyway and
//     have the issue slots -- fetch the epilogue operands of the finished tile as 16-byte row pieces in the same
//     three-steps-ahead stream as the operand loads, combine them with the LDS tile, store the rows coalesced and keep
//     the BatchNorm-backward column sums in registers (a producer wave owns 32 columns of all the tile's rows, so the
//     column sums reduce inside the wave by shuffles: deterministic, no barrier).  The epilogue of tile j thus overlaps the
//     multiply loop of tile j+1.
// Replaces F.conv2d of the 1x1 nn.Conv2d at resnet.py:33-53 (conv1, conv3), aspp.py:86-88 and their input gradients.
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"
#include "zs3hip.h"

namespace {

constexpr int PW_BN = 128;
constexpr int PW_BSTAGE = 2 * 8192;                 // weights of a K step: two 16-channel sub-chunks x 128 columns x 64 B
constexpr int PW_CTILE = 4 * PW_BN * 4;             // the epilogue's cross-wave column sums (BatchNorm partials)
constexpr int PW_LDC = PW_BN + 4;                   // floats per row of the LEPI output tile
// -DZS3_PW_ABLATE=n builds (tools/probe/build_variant.sh; timing probes, wrong results): 1 = no epilogue work (barriers kept),
// 2 = no MFMAs, 4 = no global loads, 8 = no split / LDS writes, 32 = no BatchNorm sums
#ifndef ZS3_PW_ABLATE
#define ZS3_PW_ABLATE 0
#endif
int g_pw_wgs = 256;                                 // persistent workgroups per launch (zs3_conv_pw_set_wgs)

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
// LEPI : loading epilogue run by the producer waves (header comment); Y16: its tensors (y, res, bn_y) are bf16.
template <int PREC, int BM, bool INAFF = false, bool A16 = false, bool LEPI = false, bool Y16 = false>
__global__ __launch_bounds__(512) void conv_pw_kernel(const ConvArgs p, const int ntiles, const int ntn) {
  static_assert(!A16 || (PREC == 1 && !INAFF), "bf16-stored input: plain bf16 products, no producer-side transform");
  static_assert(!LEPI || BM == 128, "the producers' epilogue tile is 128 rows");
  constexpr int BN = PW_BN, TM = BM / 64, TN = 2;
  constexpr int ASUB = BM * 64;                     // one 16-channel sub-chunk of the activation rows
  constexpr int STAGE = 2 * ASUB + PW_BSTAGE;
  constexpr int OFF_CT = 2 * STAGE;
  constexpr int OFF_EP = OFF_CT + PW_CTILE;         // (LEPI) BM x PW_LDC floats: the finished tile on its way to the producers
  constexpr int NL = PREC >= 3 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int first = xcd_remap(blockIdx.x, G);       // G <= ntiles: every workgroup owns at least one tile
  const int nmine = (ntiles - first + G - 1) / G;
  const int NK = p.cin_pad >> 5;
  const int S = nmine * NK;                         // K steps of this workgroup
  const int S3 = (S + 2) / 3 * 3;                   // the producers' schedule is unrolled by three (register sets)
  const int E = (!LEPI && p.stat_partial) ? 1 : 0;  // workgroup barriers of one (consumer-side) epilogue

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
    // load stream state: tile / K step of the next request
    int lt = first, lk = 0;
    const XT* abase = nullptr;
    unsigned rowmask = 0u;
    const unsigned char* wptr[2];
    int wstep[2];
    auto setup_tile = [&](int tile) {
      tile = tile < ntiles ? tile : first + (nmine - 1) * G;   // past the end: re-request the last tile (never multiplied)
      const int mt = tile / ntn, nt = tile - mt * ntn;
      const int m0 = mt * BM, n0 = nt * BN;
      abase = reinterpret_cast<const XT*>(p.x) + (long)(m0 + prow) * p.ldx + chan0;
      rowmask = 0u;
#pragma unroll
      for (int r = 0; r < NRA; ++r) rowmask |= (m0 + prow + RSTEP * r < p.M ? 1u : 0u) << r;
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
      f32x4 a[NRA];      // four fp32 channels, or (A16) eight bf16 channels as raw bits
      u32x4 w[2][2];
      f32x4 sc, sh;      // (INAFF) scale / shift of this lane's four channels in this K step
    };
    StepRegs buf[3];
    auto load_step = [&](StepRegs& d) {
      const bool cok = lk * 32 + chan0 < p.cin_valid;
      const XT* src = abase + lk * 32;
      if (!(ZS3_PW_ABLATE & 4)) {
#pragma unroll
        for (int r = 0; r < NRA; ++r) {
          const XT* s = (cok && ((rowmask >> r) & 1u)) ? src + r * rstep : xzero;
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

    // ---- (LEPI) the loading epilogue.  One `pass` = this wave's 32 columns x 16 rows of the finished tile; a lane owns eight
    // values of it: bf16 tensors -- eight consecutive columns of one row (one 16-byte piece of y / res / bn_y); fp32 tensors --
    // four consecutive columns of rows rip and rip + 8 (two 16-byte pieces).  The per-column BatchNorm operands of the tile sit in
    // LDS (the 2 KB the consumer-side epilogue uses for its column sums: unused here), not in registers.
    constexpr int EW = 8, CW = Y16 ? 8 : 4, NR = EW / CW, LPR = 32 / CW, NPASS = LEPI ? BM / 16 : 1;
    using YT = std::conditional_t<Y16, bf16_t, float>;
    using MT = std::conditional_t<Y16, unsigned short, unsigned char>;   // a row piece's mask bits: one byte per four columns
    const int pw = wave - 4, rip = lane / LPR, ecol = 32 * pw + (lane % LPR) * CW;   // row in pass, first column in the tile
    struct EpiRegs {
      f32x4 ro[NR], by[NR];   // raw 16-byte pieces: res or (accumulate) the old y -- never both in one launch --, and bn_y
      unsigned rm, bm;        // mask bytes of the NR pieces (piece n in bits 16 n ..)
    };
    EpiRegs ebuf[3][LEPI ? (Y16 ? 2 : 1) : 1];
    float bs_s[CW], bs_q[CW];
    float* const ccol = reinterpret_cast<float*>(dsm + OFF_CT);   // [4][128]: istd, -mean * istd, mask scale, mask shift
    int e_m0 = 0, e_n0 = 0, e_mt = 0;          // tile whose epilogue is in progress
    int e_issue = NPASS, e_cons = NPASS;       // next pass to request / to finish (NPASS: none)
    int npend[3] = {0, 0, 0};                  // passes requested into register set 0 / 1 / 2 and not finished yet
    auto epi_begin = [&](int tile) {
      const int mt = tile / ntn, nt = tile - mt * ntn;
      e_mt = mt; e_m0 = mt * BM; e_n0 = nt * BN;
      e_issue = 0; e_cons = 0;
#pragma unroll
      for (int e = 0; e < CW; ++e) {
        bs_s[e] = 0.f;
        bs_q[e] = 0.f;
      }
      if (p.bs_partial && lane < 32) {           // this wave's 32 columns; read back by the same wave only
        const int c = 32 * pw + lane, col = e_n0 + c;
        const bool cok = col < p.ncols;
        const float is = cok ? p.bs_istd[col] : 0.f;
        ccol[c] = is;
        ccol[128 + c] = cok ? -p.bs_mean[col] * is : 0.f;
        ccol[256 + c] = (cok && p.bs_msc) ? p.bs_msc[col] : 0.f;
        ccol[384 + c] = (cok && p.bs_msc) ? p.bs_msh[col] : 0.f;
      }
    };
    auto epi_issue = [&](EpiRegs& d) -> int { // request pass e_issue (if any); 1 when a request was made
      if (e_issue >= NPASS) return 0;
      const int col = e_n0 + ecol;
      const YT* zero = reinterpret_cast<const YT*>(p.zero);
      const unsigned char* zb = reinterpret_cast<const unsigned char*>(p.zero);
      const YT* rsrc = reinterpret_cast<const YT*>(p.accumulate ? p.y : p.res);
      const int ldro = p.accumulate ? p.ldy : p.ldr;
      d.rm = 0u;
      d.bm = 0u;
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        const int row = e_m0 + e_issue * 16 + rip + 8 * n;
        const bool ok = row < p.M && col < p.ncols;
        if (rsrc) d.ro[n] = *reinterpret_cast<const f32x4*>(ok ? rsrc + (size_t)row * ldro + col : zero);
        if (p.bs_partial) d.by[n] = *reinterpret_cast<const f32x4*>(ok ? reinterpret_cast<const YT*>(p.bs_y) + (size_t)row * p.bs_ldy + col : zero);
        const size_t mi = (size_t)row * (p.ncols >> 2) + (col >> 2);
        if (p.res_mbits) d.rm |= (unsigned)*reinterpret_cast<const MT*>(ok ? p.res_mbits + mi : zb) << (16 * n);
        if (p.bs_mbits) d.bm |= (unsigned)*reinterpret_cast<const MT*>(ok ? p.bs_mbits + mi : zb) << (16 * n);
      }
      ++e_issue;
      return 1;
    };
    auto unpack = [&](const f32x4 raw, float (&o)[CW]) {
      if constexpr (Y16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned u = __float_as_uint(raw[e]);
          o[2 * e] = __uint_as_float(u << 16);
          o[2 * e + 1] = __uint_as_float(u & 0xFFFF0000u);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = raw[e];
      }
    };
    auto epi_consume = [&](const EpiRegs& s) {   // finish pass e_cons with the operands requested into `s`

      const int col = e_n0 + ecol;
      const bool have_ro = p.res != nullptr || p.accumulate;
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        const int rloc = e_cons * 16 + rip + 8 * n;
        const int row = e_m0 + rloc;
        const bool ok = row < p.M && col < p.ncols;
        const float* ct = reinterpret_cast<const float*>(dsm + OFF_EP) + rloc * PW_LDC + ecol;
        float v[CW], t[CW];
#pragma unroll
        for (int e = 0; e < CW; e += 4) {
          const f32x4 c = *reinterpret_cast<const f32x4*>(ct + e);
          v[e] = c[0]; v[e + 1] = c[1]; v[e + 2] = c[2]; v[e + 3] = c[3];
        }
        if (have_ro) {
          unpack(s.ro[n], t);
          const unsigned rm = p.res_mbits ? (s.rm >> (16 * n)) : 0xFFFFFFFFu;
#pragma unroll
          for (int e = 0; e < CW; ++e) v[e] += ((rm >> (8 * (e >> 2) + (e & 3))) & 1u) ? t[e] : 0.f;   // one mask byte per four columns
        }
        if constexpr (Y16) {
          u32x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pk[e] = cvt_pk_bf16(v[2 * e], v[2 * e + 1]);
            v[2 * e] = __uint_as_float(pk[e] << 16);              // the sums below see what the next kernel will read
            v[2 * e + 1] = __uint_as_float(pk[e] & 0xFFFF0000u);
          }
          if (ok) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.y) + (size_t)row * p.ldy + col) = pk;
        } else {
          if (ok) *reinterpret_cast<f32x4*>(p.y + (size_t)row * p.ldy + col) = f32x4{v[0], v[1], v[2], v[3]};
        }
        if (p.bs_partial) {
          unpack(s.by[n], t);
          const unsigned bm = s.bm >> (16 * n);
#pragma unroll
          for (int e = 0; e < CW; ++e) {
            bool on = ok;
            if (p.bs_mbits) on = on && ((bm >> (8 * (e >> 2) + (e & 3))) & 1u);
            else if (p.bs_msc) on = on && (fmaf(t[e], ccol[256 + ecol + e], ccol[384 + ecol + e]) > 0.f);
            const float dz = on ? v[e] : 0.f;
            bs_s[e] += dz;
            bs_q[e] = fmaf(dz, fmaf(t[e], ccol[ecol + e], ccol[128 + ecol + e]), bs_q[e]);
          }
        }
      }
      if (++e_cons == NPASS && p.bs_partial) {   // the tile's column sums: rows live in lane bits log2(LPR) .. 5
#pragma unroll
        for (int e = 0; e < CW; ++e) {
#pragma unroll
          for (int o = LPR; o < 64; o <<= 1) {
            bs_s[e] += __shfl_xor(bs_s[e], o, 64);
            bs_q[e] += __shfl_xor(bs_q[e], o, 64);
          }
        }
        if (rip == 0 && col < p.ncols) {
#pragma unroll
          for (int e = 0; e < CW; ++e) {
            p.bs_partial[((size_t)e_mt * 2 + 0) * p.ncols + col + e] = bs_s[e];
            p.bs_partial[((size_t)e_mt * 2 + 1) * p.ncols + col + e] = bs_q[e];
          }
        }
      }
    };
    // passes per interval: the epilogue of a tile is requested in intervals 1 .. NK - 4 of the following tile (its LDS tile is
    // complete after the first barrier of that tile and is overwritten after the last) and finished three intervals later
    constexpr int PPI_MAX = Y16 ? 2 : 1;
    const int ppi = LEPI ? (NPASS + (NK - 4) - 1) / (NK - 4) : 0;   // <= PPI_MAX (pw_lepi_ok: NK >= 8 with bf16 tensors, >= 12 with fp32)

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
    int nexttile = NK, kloc = 0, tdone = 0;   // kloc: K step of interval i inside its tile; tdone: tiles fully multiplied before it
    for (int i0 = 0; i0 < S3; i0 += 3) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int i = i0 + r;
        write_step(buf[(r + 1) % 3], (i + 1) & 1);
        load_step(buf[(r + 1) % 3]);
        if constexpr (LEPI) {
          // (register set (r + 1) % 3: what was requested three intervals ago is finished now, then re-requested)
#pragma unroll
          for (int q = 0; q < PPI_MAX; ++q)
            if (q < npend[(r + 1) % 3]) epi_consume(ebuf[(r + 1) % 3][q]);
          npend[(r + 1) % 3] = 0;
          if (kloc == 1 && tdone >= 1 && i < S) epi_begin(first + (tdone - 1) * G);
#pragma unroll
          for (int q = 0; q < PPI_MAX; ++q)
            if (q < ppi) npend[(r + 1) % 3] += epi_issue(ebuf[(r + 1) % 3][q]);
          if (++kloc == NK) {
            kloc = 0;
            ++tdone;
          }
        }
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
    if constexpr (LEPI) {
      // passes still in flight belong to the second-to-last tile only if the padding intervals did not drain them: finish them,
      // then the last tile (its LDS tile is complete after the barrier below), two passes at a time
#pragma unroll
      for (int r = 1; r < 4; ++r)     // (issue order of the register sets: 1, 2, 0)
#pragma unroll
        for (int q = 0; q < PPI_MAX; ++q)
          if (q < npend[r % 3]) epi_consume(ebuf[r % 3][q]);
      __builtin_amdgcn_s_barrier();   // X: the consumers have written the last tile
      epi_begin(first + (nmine - 1) * G);
      for (int t = 0; t < NPASS; t += 3) {   // three passes in flight (the register sets of the pipelined form)
#pragma unroll
        for (int r = 0; r < 3; ++r) npend[r] = epi_issue(ebuf[r][0]);
#pragma unroll
        for (int r = 0; r < 3; ++r)
          if (npend[r]) epi_consume(ebuf[r][0]);
      }
    }
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
    constexpr int NMF = TM * TN * (PREC >= 3 ? 3 : 1);
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
        constexpr int ia = PREC >= 3 ? (pr == 0 ? 1 : 0) : 0, ib = PREC >= 3 ? (pr == 1 ? 1 : 0) : 0;
        if (!(ZS3_PW_ABLATE & 2))
          acc[i][j] = mfma16<PREC>(fa[SET][i][ia], fb[SET][j][ib], acc[i][j]);
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
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) unscale_acc<PREC>(acc[i][j]);   // (f16x3: the forward plane carries 2^6 w)
      if (ZS3_PW_ABLATE & 1) {
        for (int e = 0; e < E; ++e) __builtin_amdgcn_s_barrier();
        if (m0 < 0) p.y[tid] = acc[0][0][0] + acc[TM - 1][1][5];
        continue;
      }
      if constexpr (LEPI) {
        // hand the tile to the producer waves: the previous tile's LDS copy was consumed before the last barrier above
        float* ep = reinterpret_cast<float*>(dsm + OFF_EP);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
              ep[row * PW_LDC + wn * 64 + j * 32 + lr] = acc[i][j][r];
            }
        continue;   // (the next K-step barrier, or barrier X after the last tile, publishes it)
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
      if (p.y_bf16) store_acc_direct16<TM, TN, BM, BN>(p, acc, m0, n0, wm, wn, lane);
      else store_acc_direct<TM, TN, BM, BN>(p, acc, m0, n0, wm, wn, lane);
    }
    for (int e = S; e < S3; ++e) __builtin_amdgcn_s_barrier();   // the producers' schedule is padded to a multiple of three
    if constexpr (LEPI) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // X: the last tile is in LDS for the producers
    }
  }
}

// what the kernel can run at all (geometry); the epilogue / element-type questions are launch_pw's
bool pw_geom_ok(const ConvArgs& a, int bm) {
  if (a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad_h != 0 || a.pad_w != 0 || a.H != a.Ho || a.W != a.Wo) return false;
  if ((a.ldx & 3) || (a.cin_valid & 3) || (a.cin_pad & 31) || a.cin_pad < 32 || a.M <= 0) return false;
  return bm == 256 || bm == 128;
}
// loading epilogues: the producers' form (LEPI) -- 128-row tiles, >= 8 K steps per tile, vector-aligned tensors
bool pw_lepi_ok(const ConvArgs& a, int bm) {
  const int ew = a.y_bf16 ? 8 : 4;
  if (bm != 128 || (a.cin_pad >> 5) < (a.y_bf16 ? 8 : 12) || (a.res && a.accumulate) || a.stat_partial || a.scale || a.shift || a.act || a.in_scale) return false;
  if ((a.ncols % ew) || (a.ldy % ew) || (a.res && (a.ldr % ew)) || (a.bs_partial && (a.bs_ldy % ew))) return false;
  if (a.bs_partial && (!a.bs_y || !a.bs_mean || !a.bs_istd)) return false;
  if (a.res_mbits && !a.res) return false;
  return true;
}

template <int PREC, int BM, bool INAFF = false, bool A16 = false, bool LEPI = false, bool Y16 = false>
int launch_pw_t(const ConvArgs& a, hipStream_t st) {
  static bool configured = false;
  constexpr int LDS = 2 * (2 * BM * 64 + PW_BSTAGE) + PW_CTILE + (LEPI ? BM * PW_LDC * 4 : 0);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_kernel<PREC, BM, INAFF, A16, LEPI, Y16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return -4;
    configured = true;
  }
  const int ntn = (a.ncols + PW_BN - 1) / PW_BN;
  const int ntiles = ((a.M + BM - 1) / BM) * ntn;
  const int grid = ntiles < g_pw_wgs ? ntiles : g_pw_wgs;
  hipLaunchKernelGGL((conv_pw_kernel<PREC, BM, INAFF, A16, LEPI, Y16>), dim3(grid), dim3(512), LDS, st, a, ntiles, ntn);
  return ZS3_LAUNCH_CHECK();
}

}  // namespace

int zs3conv::pw_eligible(const ConvArgs& a, int bm) {
  if (!pw_geom_ok(a, bm)) return 0;
  return direct_epilogue(a) ? 1 : (pw_lepi_ok(a, bm) ? 1 : 0);
}

int zs3conv::launch_pw(const ConvArgs& a, int bm, int prec, hipStream_t st) {
  if (!pw_geom_ok(a, bm)) return -7;
  if (a.x_bf16 && (prec != 1 || (a.ldx & 7) || (a.cin_valid & 7) || a.in_scale)) return -7;
  if (a.y_bf16 && (a.ncols & 1)) return -7;
  if (!direct_epilogue(a)) {   // residual / accumulate / fused BatchNorm-backward sums: the producers' epilogue
    if (!pw_lepi_ok(a, bm)) return -7;
    if (a.x_bf16 && a.y_bf16) return launch_pw_t<1, 128, false, true, true, true>(a, st);
    if (prec == 4) return -7;   // f16x3 is the forward arithmetic; loading epilogues belong to data-gradient launches
    if (!a.x_bf16 && !a.y_bf16) return prec == 1 ? launch_pw_t<1, 128, false, false, true, false>(a, st)
                                                 : launch_pw_t<3, 128, false, false, true, false>(a, st);
    return -7;                 // mixed element types (the classifier's data gradient): the register-staged kernels
  }
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
// otherwise).  Loading epilogues additionally need tile_cfg 52 and cin_pad >= 256 (zs3_conv_igemm returns -7 otherwise).
extern "C" int zs3_conv_pw_ok(int N, int H, int W, int Ho, int Wo, int cin_pad, int cin_valid, int ldx, int KH, int KW, int stride,
                              int pad_h, int pad_w, int tile_cfg) {
  ConvArgs a{};
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.cin_pad = cin_pad; a.cin_valid = cin_valid; a.ldx = ldx;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  a.M = N * Ho * Wo;
  return pw_geom_ok(a, tile_cfg == 52 ? 128 : 256) ? 1 : 0;
}

// Persistent workgroups per launch of tile_cfg 51 / 52 (default 256 = one per CU); returns the previous value.  Tests set a
// small number so that small problems exercise the cross-tile pipeline.
extern "C" int zs3_conv_pw_set_wgs(int wgs) {
  const int old = g_pw_wgs;
  if (wgs > 0) g_pw_wgs = wgs;
  return old;
}
