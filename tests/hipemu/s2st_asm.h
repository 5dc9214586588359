// Emulator stand-in for csrc/s2st_asm.h (tests only): the reads are synchronous, the wait / fence are no-ops.
#pragma once
static inline s16x4 lds_read_tr16_raw(const unsigned char* p) { return lds_read_tr16(p); }
static inline void lds_raw_wait() {}
static inline void lds_raw_fence(s16x4&) {}
static inline void opaque_v(int&) {}
static inline void opaque_s(int&) {}
static inline void lane16_swap(unsigned& a, unsigned& b) {
  const int l = emu_lane();
  const unsigned pa = emu_shfl_idx(a, l ^ 16), pb = emu_shfl_idx(b, l ^ 16);
  const bool odd = (l >> 4) & 1;
  const unsigned na = odd ? pb : a, nb = odd ? b : pa;
  a = na;
  b = nb;
}
static inline float row16_sum(float v) {
  // the product's order of additions (quad xor 1, quad xor 2, half mirror, mirror): bit-equal sums
  const int l = emu_lane();
  v += emu_shfl_idx(v, l ^ 1);
  v += emu_shfl_idx(v, l ^ 2);
  v += emu_shfl_idx(v, (l & ~7) | (7 - (l & 7)));
  v += emu_shfl_idx(v, (l & ~15) | (15 - (l & 15)));
  return v;
}
