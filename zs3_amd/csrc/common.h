// zs3hip -- shared device helpers for the gfx950 (CDNA4 / MI355X) kernels.
// Wave size is 64 everywhere; MFMA shapes used: v_mfma_f32_32x32x16_bf16 and v_mfma_f32_32x32x2_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 accumulator fragment
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define ZS3_WAVE 64

// ---- split-bf16 ("bf16x3") arithmetic ---------------------------------------------------------
// An fp32 value x is carried through the matrix cores as hi + lo with hi = bf16(x) and
// lo = bf16(x - hi); a product is hi*hi + hi*lo + lo*hi (three bf16 MFMAs, fp32 accumulate), the
// dropped lo*lo term is <= 2^-18 |a b|.  Activations are split while they are staged into LDS (two
// v_cvt_pk_bf16_f32 per pair of values: the remainder x - hi is exact in fp32); weights are split once
// per step by zs3_prep_weight.  Both halves round to nearest even.

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {  // {bf16(a) low half, bf16(b) high half}, RNE
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned pack_hi_trunc(float a, float b) {
  return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xFFFF0000u);
}
__device__ __forceinline__ float trunc_bf16_f32(float a) { return __uint_as_float(__float_as_uint(a) & 0xFFFF0000u); }

// ---- split-fp16 ("f16x3", PREC = 4) --------------------------------------------------------------------------------------
// The same three products on v_mfma_f32_32x32x16_f16 (same rate as the bf16 instruction: 2460 vs 2375 TF in a register-only
// loop, tools/probe/f16/f16_probe.hip) with hi = fp16(x), lo = fp16(x - hi), both round-to-nearest-even (v_cvt_pk_f16_f32):
// 11 + 11 significand bits instead of 8 + 8, the dropped lo*lo term <= 2^-22 |a b| -- fp32-class products (one layer against
// fp64: 2.4e-7 .. 4.7e-7 relative, the fp32 accumulation floor, where the bf16 split gives 4.5e-6; tools/probe/fp16_split_eval.py).
// What fp16 does not have is range: lo is subnormal for |x| < 0.12 (the conversion produces and the MFMA consumes subnormals
// unflushed -- probed) and gone below |x| ~ 1e-4 (absolute error 3e-8), and |x| > 65504 overflows.  Forward operands (BatchNorm'd
// activations, O(1); weights O(1e-2)) sit inside that window, back-propagated gradients (1e-3 .. 1e-9) do not: the FORWARD
// convolutions run f16x3, data- and weight-gradient launches stay bf16x3 (exponent range of fp32).  DESIGN.md section 2.
// Weights travel through the f16x3 forward multiplied by 2^6 (exact): a convolution weight of the usual size (|w| ~ 0.03) would
// otherwise have a SUBNORMAL lo half (fp16 spacing 2^-24 below 6e-5: 1e-6 relative instead of 2^-22), which is the difference
// between 1.3x and 0.6x the reference's own error on the default-init step in the emulation (tools/probe/split_emulation.py with
// a fixed weight scale).  zs3_prep_weight_f16fwd scales, every PREC = 4 kernel multiplies its accumulators by 2^-6 (exact) before
// anything reads them.  Headroom: |w| < 1023.
#define ZS3_F16X3_WSCALE 64.0f
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {   // {fp16(a) low half, fp16(b) high half}, RNE
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// one MFMA of the 16-bit operand type PREC selects (fragments travel as 8 x 16 bit in four VGPRs either way)
template <int PREC>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (PREC == 4)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// undo the weight scale of the f16x3 forward on one accumulator fragment (no-op for every other precision)
template <int PREC>
__device__ __forceinline__ void unscale_acc(f32x16& acc) {
  if constexpr (PREC == 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] *= (1.0f / ZS3_F16X3_WSCALE);
  }
}

// lo halves from a packed hi and the two values (the second phase of a split; the LDS-DMA kernel runs the phases in different MFMA slots)
__device__ __forceinline__ unsigned f16_lo_pair(unsigned hi, float a, float b) {
  // lo = fp16(x - fp32(hi)) in ONE instruction per value: the mixed-precision FMA reads hi's halves as fp16 operands, computes
  // hi * -1.0 + x exactly (the remainder is representable in fp32) and rounds once to fp16 into the low / high half of `lo` --
  // 3 VALU per pair instead of 6 (v_cvt_f32_f16 x2, v_sub_f32 x2, v_cvt_pk_f16_f32).  Bit-identical to the long form on 4.2 M
  // pairs incl. subnormals, infinities, NaNs and arbitrary bit patterns (tools/probe/split/split_probe.hip, on the GPU).
  unsigned lo;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
  return lo;
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
// (The bf16 split has no such short form.  v_dot2_f32_bf16 with k = {-1, 0} / {0, -1} computes x - fp32(hi) in one instruction and
// is bit-identical to the long form in a flat probe kernel (tools/probe/split/split_probe.hip) -- but a DOT result needs three wait
// states before another VALU instruction reads it, which inline assembly hides from hipcc (garbage in every conv kernel), and
// the builtin form, which hipcc schedules correctly, selects the destructive v_dot2c encoding plus two copies (6 VALU again) and
// still failed the convolution tests.  Round 5; not pursued.)

template <int PREC>
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  if (PREC == 1) {
    hi = cvt_pk_bf16(a, b);
    lo = 0u;
  } else if (PREC == 4) {
    hi = cvt_pk_f16(a, b);
    lo = f16_lo_pair(hi, a, b);
  } else {
    hi = cvt_pk_bf16(a, b);  // round-to-nearest hi: |x - hi| <= 2^-9 |x|, remainder exact in fp32
    lo = cvt_pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u));
  }
}

// max(v * sc + sh, 0) on four values: two packed fp32 FMAs (v_pk_fma_f32) + four v_max -- the BatchNorm-apply + ReLU that the conv
// producers perform on their operand (ConvArgs::in_scale)
__device__ __forceinline__ f32x4 affine_relu4(f32x4 v, f32x4 sc, f32x4 sh) {
  const f32x2 lo = __builtin_elementwise_fma(f32x2{v[0], v[1]}, f32x2{sc[0], sc[1]}, f32x2{sh[0], sh[1]});
  const f32x2 hi = __builtin_elementwise_fma(f32x2{v[2], v[3]}, f32x2{sc[2], sc[3]}, f32x2{sh[2], sh[3]});
  return f32x4{fmaxf(lo[0], 0.f), fmaxf(lo[1], 0.f), fmaxf(hi[0], 0.f), fmaxf(hi[1], 0.f)};
}

// ---- bf16-STORED activations (the 2-byte mode: BASELINE configs[4], `io` arguments of the C ABI) ---------------------------
// Activation tensors are fp32 or bf16 in HBM; every kernel computes in fp32 registers.  Four channels = one 16-byte (fp32) or
// 8-byte (bf16) access per lane.  `io` bit 0 (ZS3_IO_IN16): the call's activation INPUTS are bf16, bit 1 (ZS3_IO_OUT16): its
// activation OUTPUTS are (conv: x | y, res, accumulate target, bn_y; wgrad: dy | x); strides stay in elements.
typedef unsigned short bf16_t;
#define ZS3_IO_IN16 1
#define ZS3_IO_OUT16 2
__device__ __forceinline__ f32x4 bf16x4_to_f32(u32x2 v) {
  return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xFFFF0000u), __uint_as_float(v[1] << 16),
               __uint_as_float(v[1] & 0xFFFF0000u)};
}
__device__ __forceinline__ u32x2 f32x4_to_bf16(f32x4 v);
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
// ZS3_LD_NT, defined by a source file before it includes this header: its ld4 loads are non-temporal (`global_load ... nt`) -- for
// kernels that read every element once (bn.hip's elementwise passes); files whose loads are re-read by neighbouring tiles or by the
// next kernel leave it undefined (measured per file in round 5: DESIGN.md section 7)
#ifdef ZS3_LD_NT
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) { return bf16x4_to_f32(__builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p))); }
#else
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) { return bf16x4_to_f32(*reinterpret_cast<const u32x2*>(p)); }
#endif
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return __uint_as_float(((unsigned)*p) << 16); }
// run-time element type (conv epilogues: one wave-uniform test per row, not in any inner loop); `idx` in elements
__device__ __forceinline__ f32x4 ld4_rt(const void* base, size_t idx, int bf16) {
  return bf16 ? ld4<bf16_t>(static_cast<const bf16_t*>(base) + idx) : ld4<float>(static_cast<const float*>(base) + idx);
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__device__ __forceinline__ u32x2 f32x4_to_bf16(f32x4 v) { return u32x2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])}; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f32x4 v) { *reinterpret_cast<u32x2*>(p) = f32x4_to_bf16(v); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16_rne(v); }
__device__ __forceinline__ void st4_rt(void* base, size_t idx, f32x4 v, int bf16) {
  if (bf16) st4<bf16_t>(static_cast<bf16_t*>(base) + idx, v);
  else st4<float>(static_cast<float*>(base) + idx, v);
}
__device__ __forceinline__ float ld1_rt(const void* base, size_t idx, int bf16) {
  return bf16 ? ld1<bf16_t>(static_cast<const bf16_t*>(base) + idx) : static_cast<const float*>(base)[idx];
}
__device__ __forceinline__ void st1_rt(void* base, size_t idx, float v, int bf16) {
  if (bf16) static_cast<bf16_t*>(base)[idx] = f32_to_bf16_rne(v);
  else static_cast<float*>(base)[idx] = v;
}

// XCD-aware bijective remap of a 1-D block id: blocks that land on one XCD (id % 8) get a contiguous
// chunk of logical tile ids, so tiles that share an operand panel hit the same (private) L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int nx = 8;
  int xcd = bid % nx, slot = bid / nx;
  int q = nblk / nx, r = nblk % nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// splitmix64-style counter hash -> 24-bit uniform in [0,1): the one random stream of the library (dropout masks, generator
// noise).  A value is a pure function of (seed, element index), so any kernel that needs the mask of an element recomputes it.
__device__ __forceinline__ float u01(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#define ZS3_LAUNCH_CHECK() ((int)hipGetLastError())
