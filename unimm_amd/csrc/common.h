// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the UniMM-UL hot path.
// Wave = 64 lanes everywhere; bf16 storage is uint16_t bit patterns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/unimm_hip.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

#define UNIMM_CHECK_LAUNCH()                                         \
  do {                                                               \
    hipError_t e__ = hipGetLastError();                              \
    if (e__ != hipSuccess) return UNIMM_E_HIP;                       \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even through the hardware convert (keeps NaN a NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}

// two fp32 -> packed bf16x2 in ONE v_cvt_pk_bf16_f32 (scalar casts + shift/or cost 4 VALU ops per pair)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// erf to fp32 accuracy (Abramowitz-Stegun 7.1.26, |abs err| <= 1.5e-7) in ~12 VALU ops: one v_rcp (the raw 1-ulp instruction: 1/x or __frcp_rn expand to the ~10-instruction IEEE division), one
// v_exp and a 5-term Horner chain.  The library erff costs ~4x that and dominated the GELU epilogues of
// the K=768 GEMMs (64 values per lane per tile).
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
// erf-form GELU (reference: models/vilbert_dialog.py:115-121): x * 0.5 * (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
// d/dx gelu(x) = Phi(x) + x*phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// GELU and its derivative from ONE erf / ONE exp (exp(-x^2/2) is both erf's tail and the Gaussian pdf):
// the forward FFN epilogue stores gelu'(u) instead of u, so the backward epilogue is a plain multiply.
__device__ __forceinline__ void gelu_and_grad(float x, float& y, float& dy) {
  const float t = x * 0.70710678118654752f;
  const float ax = fabsf(t);
  const float e = __expf(-ax * ax);
  const float r = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, r, -1.453152027f);
  p = fmaf(p, r, 1.421413741f);
  p = fmaf(p, r, -0.284496736f);
  p = fmaf(p, r, 0.254829592f);
  const float erfv = copysignf(1.0f - p * r * e, t);
  const float cdf = 0.5f * (1.0f + erfv);
  y = x * cdf;
  dy = fmaf(x * 0.3989422804014327f, e, cdf);
}

// Two elements at a time: the multiplies / fmas become v_pk_mul_f32 / v_pk_fma_f32 (two fp32 lanes per instruction slot);
// only |x|, exp2, rcp and the sign select stay per element.  Same formulas as gelu_and_grad (erfc(|t|) = p(r) r e^{-t^2}).
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_and_grad2(f32x2v x, f32x2v& y, f32x2v& dy) {
  const f32x2v t = x * 0.70710678118654752f;
  f32x2v ax;
  ax.x = fabsf(t.x); ax.y = fabsf(t.y);
  const f32x2v arg = (ax * ax) * -1.4426950408889634f;
  f32x2v e, r;
  e.x = __builtin_amdgcn_exp2f(arg.x); e.y = __builtin_amdgcn_exp2f(arg.y);
  const f32x2v den = ax * 0.3275911f + 1.0f;
  r.x = __builtin_amdgcn_rcpf(den.x); r.y = __builtin_amdgcn_rcpf(den.y);
  f32x2v p = r * 1.061405429f + -1.453152027f;
  p = p * r + 1.421413741f;
  p = p * r + -0.284496736f;
  p = p * r + 0.254829592f;
  const f32x2v hq = (p * r) * (e * 0.5f);                // erfc(|t|) / 2 = Phi(-|x|)
  const f32x2v om = 1.0f - hq;
  f32x2v cdf;
  cdf.x = t.x < 0.f ? hq.x : om.x; cdf.y = t.y < 0.f ? hq.y : om.y;
  y = x * cdf;
  dy = (x * 0.3989422804014327f) * e + cdf;
}

// Counter-based dropout.  Element (row, col) of a [rows, ncols] activation is kept iff the 16-bit field
// (col & 1) of  drop_word(key, row * ceil(ncols / 2) + (col >> 1))  is >= thr >> 16, key = per-(step, site)
// word built on the host (unimm_amd/dropout.py mirrors this bit for bit so the oracle can replay the
// masks).  One hash serves two neighbouring columns.  p is resolved to 2^-16.
//
// The hash is a LINEAR stage followed by one xorshift-multiply-xorshift round:
//     drop_lin(w) = w * M1 + key                      (a bijection of w for every key; affine in w)
//     drop_fin(x) = x ^= x >> 15; x ^= key * GOLD; x *= M2; x ^= x >> 16
// Affine means drop_lin(a + b) = drop_lin(a) + b * M1: a kernel forms the linear stage of the element it visits from a
// per-lane constant (computed once), a wave-uniform term (scalar unit) and compile-time constants with ONE vector add,
// and pays one 32-bit multiply (quarter rate on this ISA) per hash instead of three (index, and the two of the
// mix32-style hash of rounds 1-2: 40 % of the vector work of the attention backward kernels).  The key enters twice,
// additively and -- through another odd multiple -- by XOR before the multiply: with the addition alone the mask of key k2
// would be the mask of key k1 shifted by (k2 - k1) / M1 words.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
constexpr uint32_t DROP_M1 = 0x7feb352dU, DROP_M2 = 0x846ca68bU;
struct DropoutArg {
  uint32_t key;    // 0 with thr == 0 => disabled
  uint32_t thr;    // p * 2^32 (the kernels compare 16-bit fields against thr >> 16)
  float scale;     // 1 / (1 - p)
  const uint32_t* salt;   // or NULL: device word XORed into the key at kernel entry.  Replayed launch sequences freeze their
                          // arguments, so the per-(seed, site) key is the argument and the per-step part lives in memory.
  uint32_t key2;   // key * GOLD, filled in by drop_resolve
};
// effective key of this launch (call once at kernel entry, on the kernel's own copy of its parameters)
__device__ __forceinline__ void drop_resolve(DropoutArg& d) {
  if (d.salt != nullptr) { d.key ^= d.salt[0]; d.salt = nullptr; }
  d.key2 = d.key * 0x9E3779B1u;
}
__device__ __forceinline__ uint32_t drop_lin(const DropoutArg& d, uint32_t widx) { return widx * DROP_M1 + d.key; }
__device__ __forceinline__ uint32_t drop_fin(const DropoutArg& d, uint32_t x) {
  x ^= x >> 15;
  x ^= d.key2;
  x *= DROP_M2;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_word(const DropoutArg& d, uint32_t widx) { return drop_fin(d, drop_lin(d, widx)); }
__device__ __forceinline__ uint32_t drop_wbase(uint32_t row, uint32_t ncols, uint32_t col) { return row * ((ncols + 1u) >> 1) + (col >> 1); }
__device__ __forceinline__ bool drop_keep(const DropoutArg& d, uint32_t w, uint32_t odd) {
  return (odd ? (w >> 16) : (w & 0xffffu)) >= (d.thr >> 16);
}
__device__ __forceinline__ float drop_sel(const DropoutArg& d, uint32_t w, uint32_t odd, float v) {
  return drop_keep(d, w, odd) ? v * d.scale : 0.0f;
}
// one element
__device__ __forceinline__ float drop_apply(const DropoutArg& d, uint32_t row, uint32_t ncols, uint32_t col, float v) {
  return drop_sel(d, drop_word(d, drop_wbase(row, ncols, col)), col & 1u, v);
}
// keep bits of 8 consecutive columns col0 .. col0+7 (col0 % 8 == 0): 4 hashes
__device__ __forceinline__ uint32_t drop_bits8(const DropoutArg& d, uint32_t row, uint32_t ncols, uint32_t col0) {
  const uint32_t l0 = drop_lin(d, drop_wbase(row, ncols, col0)), t16 = d.thr >> 16;
  uint32_t bits = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t w = drop_fin(d, l0 + (uint32_t)q * DROP_M1);
    bits |= ((w & 0xffffu) >= t16 ? 1u : 0u) << (2 * q);
    bits |= ((w >> 16) >= t16 ? 1u : 0u) << (2 * q + 1);
  }
  return bits;
}

// keep bits of 4 consecutive columns col0 .. col0+3 (col0 % 4 == 0): 2 hashes (the same words drop_bits8 forms for them)
__device__ __forceinline__ uint32_t drop_bits4(const DropoutArg& d, uint32_t row, uint32_t ncols, uint32_t col0) {
  const uint32_t l0 = drop_lin(d, drop_wbase(row, ncols, col0)), t16 = d.thr >> 16;
  uint32_t bits = 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t w = drop_fin(d, l0 + (uint32_t)q * DROP_M1);
    bits |= ((w & 0xffffu) >= t16 ? 1u : 0u) << (2 * q);
    bits |= ((w >> 16) >= t16 ? 1u : 0u) << (2 * q + 1);
  }
  return bits;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch), so
// give every XCD a contiguous chunk of the logical tile order (cdna guide T1, bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
