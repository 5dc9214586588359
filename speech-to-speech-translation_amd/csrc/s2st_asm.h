// Hand-issued LDS reads for kernels whose LDS is filled by LDS-DMA (global_load_lds).
//
// hipcc cannot see that a ds_read_b64_tr_b16 BUILTIN reads a ring slot whose DMA has already been waited for: in
// front of the first transposed read of every K-step it emits `s_waitcnt vmcnt(0)`, which drains the whole prefetch
// ring (found in the ISA of every ring-kernel instantiation with a rows-contiguous operand: those ran unpipelined).
// The instruction written as inline assembly carries no such wait; the caller orders it by hand:
//     r = lds_read_tr16_raw(p) ...          issue (asynchronous: r is NOT valid yet)
//     lds_raw_wait();                       s_waitcnt lgkmcnt(0), once per batch of reads
//     lds_raw_fence(r) ...                  every later use of r is ordered behind the wait (cdna_hip_programming.md
//                                           5.4 rule 18: a register-only consumer may otherwise be hoisted above it)
// Included as <s2st_asm.h>: the wave64 emulator of tests/ ships a synchronous stand-in under the same name.
#pragma once
#include <cstdint>
#include <type_traits>

__device__ __forceinline__ s16x4 lds_read_tr16_raw(const unsigned char* p) {
  s16x4 r;
  const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(p);
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(a));
  return r;
}
// (no "memory" clobber: the reads' results are ordered by the register fences below, and a clobber makes hipcc re-load
// kernel-argument fields -- s_load + its own lgkmcnt wait -- in every K-step)
__device__ __forceinline__ void lds_raw_wait() { asm volatile("s_waitcnt lgkmcnt(0)"); }
__device__ __forceinline__ void lds_raw_fence(s16x4& x) { asm volatile("" : "+v"(x)); }

// Opaque copies: the value is unchanged, but the compiler may not look through the statement -- work that depends on it
// cannot be hoisted above it (gemm_bf16_p4.hip pins epilogue address arithmetic behind its K-loop and keeps loop-invariant
// DMA row addresses from being materialised in registers the accumulators need).
__device__ __forceinline__ void opaque_v(int& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void opaque_s(int& x) { asm volatile("" : "+s"(x)); }

// v_permlane16_swap_b32 (gfx950): the odd 16-lane rows of `a` trade places with the even rows of `b` --
//   a' = {a row 0, b row 0, a row 2, b row 2},  b' = {a row 1, b row 1, a row 3, b row 3}
// (lane map checked on hardware: tools/ubench/permlane16_swap.hip).  One VALU instruction, no LDS crossbar.
__device__ __forceinline__ void lane16_swap(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

// Sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane of the row: four VALU adds
// with DPP operands (quad_perm xor 1, quad_perm xor 2, row_half_mirror, row_mirror) -- no LDS crossbar, no waits.
// (__shfl_xor lowers to ds_bpermute_b32 + s_waitcnt: ~4x the issue cost per step and an lgkmcnt dependency.)
__device__ __forceinline__ float row16_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1, 0, 3, 2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2, 3, 0, 1]
  v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror: lane i <-> 7 - i of each 8
  v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror: lane i <-> 15 - i
  return v;
}
