// numpy's legacy generator on the HOST, several threads.
//
// The reference draws its initial Griffin-Lim phases with np.random.rand from numpy's GLOBAL generator
// (fairseq/models/text_to_speech/vocoder.py:101-102): MT19937 (Matsumoto & Nishimura 1998; numpy/random/mtrand: 624-word
// key + position), one double per TWO 32-bit outputs: (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53.  numpy produces ~0.37 G
// doubles per second on one core; 64 utterances need 45 M.  The stream is one sequence, but its recurrence alone (no
// tempering, no conversion, no stores) runs an order of magnitude faster than the full draw, so thread t first SKIPS to
// its share's start state by regeneration only and then produces its share: the doubles, in numpy's order, bit for bit,
// from numpy's own state, plus the states at the share boundaries (so that the caller can put numpy's global generator
// where any smaller number of draws would have left it).  tests/test_inference.py compares with numpy draw for draw.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "s2st_ops.h"

namespace {

// the next 624 words (numpy: mt19937_gen).  Two-array form so that the compiler can vectorise: the in-place loop's
// dependences -- k[kk + 1] old, k[kk - 227] new -- are distance 227 here.  Built for the vector widths the host may have
// (the loader picks one: the recurrence is pure 32-bit integer work).
#define S2ST_TWIST(a, b) ((((a) & 0x80000000u) | ((b) & 0x7fffffffu)) >> 1) ^ ((0u - ((b) & 1u)) & 0x9908b0dfu)
#if defined(__HIP_DEVICE_COMPILE__)  // (host code: the device pass of the same translation unit has no x86 variants)
#define S2ST_HOST_CLONES
#else
#define S2ST_HOST_CLONES __attribute__((target_clones("avx2", "default")))
#endif
S2ST_HOST_CLONES void mt_regenerate(uint32_t* __restrict__ key) {
  alignas(64) uint32_t nw[624];
  const uint32_t* __restrict__ o = key;
  for (int kk = 0; kk < 227; ++kk) nw[kk] = o[kk + 397] ^ S2ST_TWIST(o[kk], o[kk + 1]);
  for (int kk = 227; kk < 454; ++kk) nw[kk] = nw[kk - 227] ^ S2ST_TWIST(o[kk], o[kk + 1]);
  for (int kk = 454; kk < 623; ++kk) nw[kk] = nw[kk - 227] ^ S2ST_TWIST(o[kk], o[kk + 1]);
  nw[623] = nw[396] ^ S2ST_TWIST(o[623], nw[0]);
  memcpy(key, nw, sizeof(nw));
}

struct MT {
  alignas(64) uint32_t key[624];
  int pos;

  void regenerate() {
    mt_regenerate(key);
    pos = 0;
  }
  // advance by `words` outputs without producing them
  void skip(int64_t words) {
    while (words > 0) {
      if (pos >= 624) regenerate();
      const int64_t take = words < 624 - pos ? words : 624 - pos;
      pos += (int)take;
      words -= take;
    }
  }
  static inline uint32_t temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  inline uint32_t next() {
    if (pos >= 624) regenerate();
    return temper(key[pos++]);
  }
  void doubles(double* out, int64_t n) {
    int64_t i = 0;
    while (i < n) {
      // whole pairs out of the current block in one tight loop
      if (pos >= 624) regenerate();
      const int64_t pairs = (624 - pos) / 2;
      const int64_t m = pairs < n - i ? pairs : n - i;
      const uint32_t* k = key + pos;
      for (int64_t j = 0; j < m; ++j) {
        const uint32_t a = temper(k[2 * j]) >> 5, b = temper(k[2 * j + 1]) >> 6;
        out[i + j] = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
      }
      pos += (int)(2 * m);
      i += m;
      if (i < n && pos == 623) {  // a double that straddles two blocks
        const uint32_t a = next() >> 5, b = next() >> 6;
        out[i++] = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
      }
    }
  }
};

}  // namespace

extern "C" {

// state: 624 key words + position (np.random.get_state()[1], [2]).  out: n doubles (may be NULL: states only).
// bounds_out (optional): (threads + 1) records of 625 words: the generator's state in front of double n * t / threads
// (t = 0 .. threads; the last one = after all n).  Returns 0.
int s2st_mt19937_host_doubles(const uint32_t* state, int64_t n, double* out, uint32_t* bounds_out, int32_t threads) {
  if (!state || n < 0 || threads < 1) return S2ST_ERR_ARG;
  if (threads > 64) threads = 64;
  std::vector<std::thread> th;
  auto work = [&](int t) {
    MT g;
    memcpy(g.key, state, sizeof(g.key));
    g.pos = (int)state[624];
    const int64_t d0 = n * t / threads, d1 = n * (t + 1) / threads;
    g.skip(2 * d0);
    if (bounds_out) {
      memcpy(bounds_out + (int64_t)t * 625, g.key, sizeof(g.key));
      bounds_out[(int64_t)t * 625 + 624] = (uint32_t)g.pos;
    }
    if (out) g.doubles(out + d0, d1 - d0);
    else g.skip(2 * (d1 - d0));
    if (bounds_out && t == threads - 1) {
      memcpy(bounds_out + (int64_t)threads * 625, g.key, sizeof(g.key));
      bounds_out[(int64_t)threads * 625 + 624] = (uint32_t)g.pos;
    }
  };
  for (int t = 1; t < threads; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  return 0;
}

}  // extern "C"
