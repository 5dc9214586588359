// Device-side helpers shared by the s2st HIP kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "s2st_hip.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));
typedef float f32v4 __attribute__((ext_vector_type(4)));

#define S2ST_WAVE 64

// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q, columns
// 4p..4p+3 of a 4 x 16 block of 16-bit elements; lane i receives column i, row q in element q
// (cdna_hip_programming.md T10).  Used to read K-strided ("rows-contiguous") GEMM operands
// from an LDS image kept in its natural [k][rows] layout.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_read_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

#include <s2st_asm.h>  // lds_read_tr16_raw / lds_raw_wait / lds_raw_fence (angle brackets: the test emulator shadows it)

// 4 x f32 -> 4 x bf16 (round-to-nearest-even), packed in 8 bytes.
// Lowers to two v_cvt_pk_bf16_f32 on gfx950.
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) {
  f32v4 f = {a, b, c, d};
  bf16v4 h = __builtin_convertvector(f, bf16v4);
  union { bf16v4 v; uint2 u; } cv;
  cv.v = h;
  return cv.u;
}

// hi/lo split: x ~= hi + lo with both bf16 (error ~2^-17 |x|): the "bf16x3" precise mode.
__device__ __forceinline__ void split_bf16x4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
  f32v4 f = {a, b, c, d};
  bf16v4 h = __builtin_convertvector(f, bf16v4);
  f32v4 hf = __builtin_convertvector(h, f32v4);
  f32v4 r = f - hf;
  bf16v4 l = __builtin_convertvector(r, bf16v4);
  union { bf16v4 v; uint2 u; } c1, c2;
  c1.v = h;
  c2.v = l;
  hi = c1.u;
  lo = c2.u;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// exact GELU (nn.GELU(), fairseq utils.gelu: 0.5 x (1 + erf(x / sqrt 2)))
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
// The same function for the bf16-operand kernels' epilogues (HuBERT's conv stack and FFN: 670 M evaluations per 24 x 8 s
// batch, where libm's branchy erff was ~0.6 ms of a 6.9 ms forward): erfc(|z|) = t (a1 + t (a2 + t (a3 + t (a4 + t a5))))
// exp(-z^2), t = 1 / (1 + p |z|) (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 -- fp32's own epsilon; results are rounded
// to bf16 or feed bf16 operands), composed so that the negative tail has no 1 - erf cancellation:
// gelu(x) = x >= 0 ? x - 0.5 x erfc(z) : 0.5 x erfc(-z).  One v_rcp, one v_exp, eight multiply-adds, no branch.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float az = fabsf(x) * 0.70710678118654752f;
  const float t = __frcp_rn(1.f + 0.3275911f * az);
  const float e = __expf(-az * az);
  float q = 1.061405429f;
  q = q * t - 1.453152027f;
  q = q * t + 1.421413741f;
  q = q * t - 0.284496736f;
  q = q * t + 0.254829592f;
  q = q * t * e;                 // erfc(|z|)
  const float r = 0.5f * x * q;  // sign of x
  return x >= 0.f ? x - r : r;
}

// Counter-based dropout RNG: keep(element) is a pure function of (seed, element index), so
// the backward pass regenerates the forward mask without storing it.
// Round 3: a 32-bit mixer with TWO 32-bit multiplies (the "lowbias32" constants; the seed's halves enter before the first
// and between the two rounds, so two sites' streams are not index-shifted copies of each other).  Rounds 1 - 2 used a
// 64-bit finalizer: three 64 x 64 multiplies = ~12 quarter-rate v_mul per element, which clock stamps showed to be the
// largest single cost of the attention kernels (8 elements per lane and key tile: ~1.6 k of the ~3.8 k cycles a wave
// spent per tile) and of every GEMM epilogue with dropout (64 elements per lane of a 128 x 128 tile).
// Round 4: the seed is whitened first (one 64-bit multiply + xor-shift on a wave-uniform value: scalar instructions, hoisted
// out of the element loops), so that BOTH words the rounds consume depend on every bit of the seed.  Without it two seeds
// that differ only in their low word gave XOR-permuted copies of one mask (mask_s'(i) = mask_s(i ^ (s ^ s')): found by
// tests/test_ops.py::test_dropout_mask_independence, phi = 0.5 at lag 1 for seed + 1); the engine's own site seeds always
// differ in both words, so no training run was affected.
__device__ __forceinline__ uint32_t mix32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ull;
  z ^= z >> 32;
  const uint32_t hi = (uint32_t)(idx >> 32);
  uint32_t h = (uint32_t)idx ^ (uint32_t)z ^ ((hi << 16) | (hi >> 16));
  h ^= h >> 16;
  h *= 0x7feb352du;
  h ^= h >> 15;
  h ^= (uint32_t)(z >> 32);
  h *= 0x846ca68bu;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  // returns 0 (dropped) or 1/(1-p) (kept)
  const uint32_t h = mix32(seed, idx);
  float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.0f;
}

// "slow index" -> element offset with an optional split (used for [B][T(+halo)][C]
// buffers addressed by a flat row id): off(i) = (i / per) * bs + (i % per) * ld
typedef s2st_split Split;
__device__ __forceinline__ long split_off(const Split& s, int i) {
  if (s.per <= 0) return (long)i * s.ld;
  int q = i / s.per;
  return (long)q * s.bs + (long)(i - q * s.per) * s.ld;
}
