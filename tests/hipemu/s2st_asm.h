// Emulator stand-in for csrc/s2st_asm.h (tests only): the reads are synchronous, the wait / fence are no-ops.
#pragma once
static inline s16x4 lds_read_tr16_raw(const unsigned char* p) { return lds_read_tr16(p); }
static inline void lds_raw_wait() {}
static inline void lds_raw_fence(s16x4&) {}
