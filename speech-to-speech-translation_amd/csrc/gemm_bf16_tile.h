// Device-side building blocks shared by the bf16 GEMM kernels (gemm_bf16.hip, gemm_bf16_w4.hip): LDS tile images and
// their MFMA fragment readers (Stage), the LDS-DMA loaders (Dma), and the epilogues.  Everything lives in an anonymous
// namespace: each translation unit gets its own copy.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "s2st_ops.h"
#include "s2st_prof.h"

namespace {

typedef unsigned short bf16_t;
constexpr int BK = 64;

template <int ROWS>
__device__ __forceinline__ int kperm(int kr) {
  // XOR key (in 32-byte pairs) of k row kr of a rows-contiguous image
  if (ROWS >= 128) return (kr & 3) | (((kr >> 3) & 1) << 2);
  return ((kr >> 1) & 1) | (((kr >> 3) & 1) << 1);
}

__device__ __forceinline__ uint4 mask_tail(uint4 v, int nvalid) {
  // keep the first nvalid (0..8) bf16 elements of a 16-byte chunk, zero the rest
  unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int e = nvalid - 2 * i;
    w[i] = e >= 2 ? w[i] : (e == 1 ? (w[i] & 0xffffu) : 0u);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <bool KM, int ROWS, bool VEC>
struct Stage {
  static constexpr int NCH = ROWS * 8 / 256;  // 16-byte chunks per thread per K step
  static constexpr int RC = ROWS / 8;         // chunks per k row (rows-contiguous image)
  static constexpr int PITCH = KM ? 128 : ROWS * 2;
  static constexpr int BYTES = ROWS * 128;
  static constexpr int READS_PER_FRAG = KM ? 1 : 2;  // LDS read instructions behind one frag()
  uint4 r[NCH];
  long off[NCH];  // KM: element offset of this chunk's row (clamped) ; !KM: clamped first row
  int tid;

  __device__ __forceinline__ void init(const GemmOperand& X, int r0, int R, int tid_) {
    tid = tid_;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int f = tid + 256 * i;
      if (KM) {
        const int row = r0 + (f >> 3);
        if (VEC) off[i] = split_off(X.sp, min(row, R - 1));
        else off[i] = row < R ? split_off(X.sp, row) : -1;
      } else {
        const int rr = r0 + (f % RC) * 8;
        // VEC: rows are readable up to the next multiple of 8 (launcher contract)
        off[i] = VEC ? min(rr, ((R + 7) & ~7) - 8) : rr;
      }
    }
  }

  __device__ __forceinline__ void load(const GemmOperand& X, const bf16_t* base, int R, int kt, int kend) {
    const bool tail = kt + BK > kend;  // wave-uniform: only the last K step of a chunk
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int f = tid + 256 * i;
      if (KM) {
        const int k0 = kt + (f & 7) * 8;
        if (VEC) {
          if (!tail) {
            r[i] = *reinterpret_cast<const uint4*>(base + off[i] + k0);
          } else {
            const int nv = kend - k0;  // valid elements of this chunk
            uint4 v = make_uint4(0, 0, 0, 0);
            if (nv > 0) v = *reinterpret_cast<const uint4*>(base + off[i] + k0);
            r[i] = nv >= 8 ? v : mask_tail(v, nv > 0 ? nv : 0);
          }
        } else {
          unsigned short e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = (off[i] >= 0 && k0 + j < kend) ? base[off[i] + k0 + j] : 0;
          r[i] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        }
      } else {
        const int k = kt + f / RC;
        const bool ok = k < kend;
        const long o = split_off(X.sp, min(k, kend - 1));
        if (VEC) {
          uint4 v = *reinterpret_cast<const uint4*>(base + o + off[i]);
          r[i] = ok ? v : make_uint4(0, 0, 0, 0);
        } else {
          unsigned short e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = (ok && off[i] + j < R) ? base[o + off[i] + j] : 0;
          r[i] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        }
      }
    }
  }

  __device__ __forceinline__ void store(unsigned char* img) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int f = tid + 256 * i;
      int o;
      if (KM) {
        const int row = f >> 3, ch = f & 7;
        o = row * 128 + ((ch ^ (row & 7)) << 4);
      } else {
        const int kr = f / RC, ch = f % RC;
        o = kr * PITCH + (((((ch >> 1) ^ kperm<ROWS>(kr)) << 1) | (ch & 1)) << 4);
      }
      *reinterpret_cast<uint4*>(img + o) = r[i];
    }
  }

  // The same fragment in two steps for the LDS-DMA kernels (s2st_asm.h): raw() issues the read(s) -- transposed reads
  // by hand, without the compiler's vmcnt(0) -- and done() builds the operand after the caller's lds_raw_wait().
  struct Raw { bf16x8 v; s16x4 lo, hi; };
  __device__ static __forceinline__ Raw raw(const unsigned char* img, int rt, int s, int lane) {
    Raw r;
    if (KM) {
      r.v = frag(img, rt, s, lane);
    } else {
      const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
      const int kr = 32 * s + 8 * g + q;
      const unsigned char* a = img + kr * PITCH + ((((rt >> 4) ^ kperm<ROWS>(kr))) << 5) + 8 * p;
      r.lo = lds_read_tr16_raw(a);
      r.hi = lds_read_tr16_raw(a + 4 * PITCH);  // k rows +4: same XOR key
    }
    return r;
  }
  __device__ static __forceinline__ bf16x8 done(Raw& r) {
    if (KM) return r.v;
    lds_raw_fence(r.lo);
    lds_raw_fence(r.hi);
    bf16x8 fr;
    fr[0] = r.lo[0]; fr[1] = r.lo[1]; fr[2] = r.lo[2]; fr[3] = r.lo[3];
    fr[4] = r.hi[0]; fr[5] = r.hi[1]; fr[6] = r.hi[2]; fr[7] = r.hi[3];
    return fr;
  }

  // MFMA 16x16x32 operand fragment, tile rows [rt, rt+16) (rt % 16 == 0), k step s (0/1):
  // lane l holds k = 32 s + 8 (l >> 4) + j
  __device__ static __forceinline__ bf16x8 frag(const unsigned char* img, int rt, int s, int lane) {
    if (KM) {
      const int row = rt + (lane & 15), ch = 4 * s + (lane >> 4);
      return *reinterpret_cast<const bf16x8*>(img + row * 128 + ((ch ^ (row & 7)) << 4));
    } else {
      const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
      const int kr = 32 * s + 8 * g + q;
      const unsigned char* a = img + kr * PITCH + ((((rt >> 4) ^ kperm<ROWS>(kr))) << 5) + 8 * p;
      s16x4 lo = lds_read_tr16(a);
      s16x4 hi = lds_read_tr16(a + 4 * PITCH);  // k rows +4: same XOR key
      bf16x8 fr;
      fr[0] = lo[0]; fr[1] = lo[1]; fr[2] = lo[2]; fr[3] = lo[3];
      fr[4] = hi[0]; fr[5] = hi[1]; fr[6] = hi[2]; fr[7] = hi[3];
      return fr;
    }
  }
};

// Straight-line epilogue for the common case (16-byte aligned outputs, N % 4 == 0, no split-K, no output mask, no GELU):
// every wave-uniform choice is a template parameter or folded into arithmetic (bias = a zero vector when absent, ReLU =
// max with 0 or -inf), so the eight accumulator blocks of a lane cost their loads, a few VALU operations and their
// stores.  The general epilogue below tests ~12 kernel-argument conditions per block: measured with in-kernel clock
// stamps (tools/gemm_stamp.sh) it took 5.3 k cycles per 128 x 128 tile with a bf16 output alone and 15.9 k with bias +
// residual -- as long as the whole K-loop of a K = 512 product (7.6 k).
template <int BM, int BN, int WGN, bool DROP, bool RESID, bool ACC, bool HASC, bool HASH, bool H16, bool GELU = false>
__device__ __forceinline__ void gemm_epilogue_fast(const GemmArgs& g, f32x4 (&acc)[BM / 32][BN / (16 * WGN)], int m0, int n0,
                                                   int wm, int wn, int lane, int zb, int zq, int zr,
                                                   const float4 (*pre)[BN / (16 * WGN)]) {
  constexpr int WM = BM / 2, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  const long czoff = zq * g.C.zo + zr * g.C.zi;
  float* cbase = g.C.p + czoff;
  bf16_t* hbase = g.C.h + czoff;
  const float* rbase = g.ep.resid + czoff;
  const int M = g.M, N = g.N;
  const float alpha = g.ep.alpha, lo = g.ep.act == 1 ? 0.f : -__builtin_inff();
  const float drop_p = g.ep.drop_p, inv_keep = DROP ? 1.f / (1.f - drop_p) : 1.f;
  const uint64_t seed = g.ep.seed;
  int ncol[TN];
  bool nok[TN];
  float4 b4[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
    nok[j] = n < N;
    ncol[j] = nok[j] ? n : N - 4;  // (N % 4 == 0, N >= 4: a valid, aligned column group for the loads of masked lanes)
    b4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (g.ep.bias) {
    const float* bias = g.ep.bias + zq * g.ep.bias_zo;
#pragma unroll
    for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const float4*>(bias + ncol[j]);
  }
  long roff[TM];
  int mrow[TM];
  bool mok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WM + i * 16 + (lane & 15);
    mok[i] = m < M;
    mrow[i] = mok[i] ? m : M - 1;
    roff[i] = split_off(g.C.sp, mrow[i]);
  }
  // all loads of the tile first (residual / old value), then arithmetic and stores
  float4 r4[RESID ? TM : 1][RESID ? TN : 1], c4[ACC ? TM : 1][ACC ? TN : 1];
  // (pre: optionally, the same values already in registers.  Fetching them before the K-loop was tried: the 64 KB per
  // workgroup delay the ring's prologue by ~6 k cycles and the step got slower, 9.28 vs 8.98 ms)
  if (RESID) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        r4[RESID ? i : 0][RESID ? j : 0] = pre ? pre[i][j] : *reinterpret_cast<const float4*>(rbase + roff[i] + ncol[j]);
  }
  if (ACC) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        c4[ACC ? i : 0][ACC ? j : 0] = pre ? pre[i][j] : *reinterpret_cast<const float4*>(cbase + roff[i] + ncol[j]);
  }
  uint2 hp[(HASH && H16) ? TM : 1][(HASH && H16) ? TN : 1];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float v[4] = {fmaxf(alpha * acc[i][j][0] + b4[j].x, lo), fmaxf(alpha * acc[i][j][1] + b4[j].y, lo),
                    fmaxf(alpha * acc[i][j][2] + b4[j].z, lo), fmaxf(alpha * acc[i][j][3] + b4[j].w, lo)};
      if (GELU) {  // (act == 2: lo is -inf)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(v[r]);
      }
      if (DROP) {
        const uint64_t e0 = ((uint64_t)zb * M + mrow[i]) * (uint64_t)N + ncol[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= drop_scale(seed, e0 + r, drop_p, inv_keep);
      }
      if (RESID) {
        const float4 t = r4[RESID ? i : 0][RESID ? j : 0];
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      }
      if (ACC) {
        const float4 t = c4[ACC ? i : 0][ACC ? j : 0];
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      }
      if (mok[i] && nok[j]) {
        if (HASC) *reinterpret_cast<float4*>(cbase + roff[i] + ncol[j]) = make_float4(v[0], v[1], v[2], v[3]);
        if (HASH && !H16) *reinterpret_cast<uint2*>(hbase + roff[i] + ncol[j]) = pack_bf16x4(v[0], v[1], v[2], v[3]);
      }
      if (HASH && H16) hp[(HASH && H16) ? i : 0][(HASH && H16) ? j : 0] = pack_bf16x4(v[0], v[1], v[2], v[3]);
    }
  }
  if (HASH && H16) {
    // 16-byte bf16 stores (the store path moves ~7 B/clk/CU with 8-byte pieces, about twice that with 16-byte ones):
    // lanes l and l ^ 16 hold columns 4g .. 4g+3 and 4g+4 .. 4g+7 of the same rows; for a pair of row blocks (i0, i1)
    // one v_permlane16_swap per dword leaves the even groups with 8 columns of block i0 and the odd groups with 8
    // columns of block i1
    const bool odd = (lane >> 4) & 1;
#pragma unroll
    for (int i = 0; i + 1 < TM; i += 2) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        unsigned ax = hp[i][j].x, ay = hp[i][j].y, bx = hp[i + 1][j].x, by = hp[i + 1][j].y;
        lane16_swap(ax, bx);
        lane16_swap(ay, by);
        const int n = n0 + wn * WN + j * 16 + ((lane >> 4) & 2) * 4;  // first of the 8 columns
        const bool ok = (odd ? mok[i + 1] : mok[i]) && n < N;
        const long ro = odd ? roff[i + 1] : roff[i];
        if (ok) *reinterpret_cast<uint4*>(hbase + ro + n) = make_uint4(ax, ay, bx, by);
      }
    }
  }
}

// The backward of ReLU + dropout folded into a data-gradient product (dX_pre = mask(dY W) from the layer's bf16 output
// y: zero where y == 0, scaled elsewhere), bf16 result only, plus the bias gradient (column sums, fp32 atomics): the
// straight-line form of that case.
template <int BM, int BN, int WGN>
__device__ __forceinline__ void gemm_epilogue_fast_masky(const GemmArgs& g, f32x4 (&acc)[BM / 32][BN / (16 * WGN)], int m0, int n0,
                                                         int wm, int wn, int lane, int zq, int zr) {
  constexpr int WM = BM / 2, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  const long czoff = zq * g.C.zo + zr * g.C.zi;
  bf16_t* hbase = g.C.h + czoff;
  const bf16_t* ybase = g.ep.mask_y + czoff;
  const int M = g.M, N = g.N;
  const float alpha = g.ep.alpha * g.ep.mask_scale;
  int ncol[TN];
  bool nok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
    nok[j] = n < N;
    ncol[j] = nok[j] ? n : N - 4;
  }
  long roff[TM];
  bool mok[TM];
  uint2 y4[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WM + i * 16 + (lane & 15);
    mok[i] = m < M;
    roff[i] = split_off(g.C.sp, mok[i] ? m : M - 1);
#pragma unroll
    for (int j = 0; j < TN; ++j) y4[i][j] = *reinterpret_cast<const uint2*>(ybase + roff[i] + ncol[j]);
  }
  float cs[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j) cs[j][0] = cs[j][1] = cs[j][2] = cs[j][3] = 0.f;
  uint2 hp[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const unsigned yw[4] = {y4[i][j].x & 0x7fffu, (y4[i][j].x >> 16) & 0x7fffu, y4[i][j].y & 0x7fffu, (y4[i][j].y >> 16) & 0x7fffu};
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = yw[r] != 0 ? alpha * acc[i][j][r] : 0.f;  // +-0 -> no gradient
        if (mok[i] && nok[j]) cs[j][r] += v[r];
      }
      hp[i][j] = pack_bf16x4(v[0], v[1], v[2], v[3]);
    }
  }
  const bool odd = (lane >> 4) & 1;
#pragma unroll
  for (int i = 0; i + 1 < TM; i += 2) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      unsigned ax = hp[i][j].x, ay = hp[i][j].y, bx = hp[i + 1][j].x, by = hp[i + 1][j].y;
      lane16_swap(ax, bx);
      lane16_swap(ay, by);
      const int n = n0 + wn * WN + j * 16 + ((lane >> 4) & 2) * 4;
      const bool ok = (odd ? mok[i + 1] : mok[i]) && n < N;
      const long ro = odd ? roff[i + 1] : roff[i];
      if (ok) *reinterpret_cast<uint4*>(hbase + ro + n) = make_uint4(ax, ay, bx, by);
    }
  }
  if (g.ep.colsum || g.ep.colsum_part) {
    // column sums over the wave's rows (the 16 lanes of a DPP row hold 16 rows of a column group): DPP row sums, then
    // either this (row tile, wave row)'s partial row -- folded in order by the caller, no atomics -- or atomic adds
    float* prow = g.ep.colsum_part ? g.ep.colsum_part + ((long)(m0 / BM) * 2 + wm) * N : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float t[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] = row16_sum(cs[j][r]);
      if ((lane & 15) == 0 && nok[j]) {
        if (prow) *reinterpret_cast<float4*>(prow + ncol[j]) = make_float4(t[0], t[1], t[2], t[3]);
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(g.ep.colsum + ncol[j] + r, t[r]);
        }
      }
    }
  }
}

// output forms that exist: fp32 only; bf16 only / fp32 + bf16 with the 16-byte bf16 stores (the launcher marks a product
// fast only if its bf16 copy qualifies for them); accumulate only into an fp32-only output (weight gradients)
template <int BM, int BN, int WGN, bool DROP, bool RESID, bool ACC, bool GELU = false>
__device__ __forceinline__ void gemm_epilogue_fast_out(const GemmArgs& g, f32x4 (&acc)[BM / 32][BN / (16 * WGN)], int m0, int n0,
                                                       int wm, int wn, int lane, int zb, int zq, int zr,
                                                       const float4 (*pre)[BN / (16 * WGN)]) {
  constexpr bool PAIRS = (BM / 32) % 2 == 0;  // row blocks per wave come in pairs (every tile shape in use)
  if constexpr (ACC) {
    gemm_epilogue_fast<BM, BN, WGN, DROP, RESID, ACC, true, false, false>(g, acc, m0, n0, wm, wn, lane, zb, zq, zr, pre);
  } else {
    if (g.C.p && g.C.h) gemm_epilogue_fast<BM, BN, WGN, DROP, RESID, ACC, true, true, PAIRS, GELU>(g, acc, m0, n0, wm, wn, lane, zb, zq, zr, pre);
    else if (g.C.p) gemm_epilogue_fast<BM, BN, WGN, DROP, RESID, ACC, true, false, false, GELU>(g, acc, m0, n0, wm, wn, lane, zb, zq, zr, pre);
    else gemm_epilogue_fast<BM, BN, WGN, DROP, RESID, ACC, false, true, PAIRS, GELU>(g, acc, m0, n0, wm, wn, lane, zb, zq, zr, pre);
  }
}

// Epilogue shared by the kernels.  acc holds C^T tiles: lane -> m = lane & 15, n = 4 (lane >> 4) + r.  FAST: also carry
// the straight-line forms above (the one-shot / grouped ring kernels; the other kernels keep compile time down without).
template <int BM, int BN, int WGN, bool FAST = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[BM / 32][BN / (16 * WGN)], int m0,
                                              int n0, int wm, int wn, int lane, int zb, int ks, int zq, int zr,
                                              const float4 (*pre)[BN / (16 * WGN)] = nullptr) {
  constexpr int WM = BM / 2, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  const long czoff = zq * g.C.zo + zr * g.C.zi;
  float* cbase = g.C.p ? g.C.p + czoff : nullptr;
  bf16_t* hbase = g.C.h ? g.C.h + czoff : nullptr;
  const float* rbase = g.ep.resid ? g.ep.resid + czoff : nullptr;
  const float inv_keep = g.ep.drop_p > 0.f ? 1.f / (1.f - g.ep.drop_p) : 1.f;
  const bool lead = (ks == 0);
  const bool cvec = g.cvec != 0;
  if constexpr (FAST) {
  if (g.cvec & 8) {  // (launcher: the masked bf16-only form, 16-byte stores)
    if constexpr ((BM / 32) % 2 == 0) {
      gemm_epilogue_fast_masky<BM, BN, WGN>(g, acc, m0, n0, wm, wn, lane, zq, zr);
      return;
    }
  }
  if (g.cvec & 2) {  // (launcher, mark_fast_epilogue: aligned, N % 4 == 0, no split-K / slab / output mask / column sums / GELU)
    const bool drop = g.ep.drop_p > 0.f, res = g.ep.resid != nullptr, accu = g.ep.accumulate != 0;
    if (g.ep.act == 2) {  // (launcher: GELU only without dropout / accumulation -- the frozen HuBERT front end's products)
      if (res) gemm_epilogue_fast_out<BM, BN, WGN, false, true, false, true>(g, acc, m0, n0, wm, wn, lane, zb, zq, zr, pre);
      else gemm_epilogue_fast_out<BM, BN, WGN, false, false, false, true>(g, acc, m0, n0, wm, wn, lane, zb, zq, zr, pre);
    } else
    if (!drop && !res && !accu) gemm_epilogue_fast_out<BM, BN, WGN, false, false, false>(g, acc, m0, n0, wm, wn, lane, zb, zq, zr, pre);
    else if (!drop && res && !accu) gemm_epilogue_fast_out<BM, BN, WGN, false, true, false>(g, acc, m0, n0, wm, wn, lane, zb, zq, zr, pre);
    else if (!drop && !res && accu) gemm_epilogue_fast_out<BM, BN, WGN, false, false, true>(g, acc, m0, n0, wm, wn, lane, zb, zq, zr, pre);
    else if (drop && !res && !accu) gemm_epilogue_fast_out<BM, BN, WGN, true, false, false>(g, acc, m0, n0, wm, wn, lane, zb, zq, zr, pre);
    else gemm_epilogue_fast_out<BM, BN, WGN, true, true, false>(g, acc, m0, n0, wm, wn, lane, zb, zq, zr, pre);  // drop && res && !accu
    return;
  }
  }
  if (g.slab) {
    // split-K partial: alpha * acc into slab[blockIdx.y][M][N] (dense); splitk_reduce_kernel combines
    float* sb = g.slab + (long)blockIdx.y * g.M * g.N;
    const bool v4 = (g.N & 3) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WM + i * 16 + (lane & 15);
      if (m >= g.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
        if (n >= g.N) continue;
        float* dst = sb + (long)m * g.N + n;
        if (v4) {
          *reinterpret_cast<float4*>(dst) = make_float4(g.ep.alpha * acc[i][j][0], g.ep.alpha * acc[i][j][1],
                                                        g.ep.alpha * acc[i][j][2], g.ep.alpha * acc[i][j][3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < g.N) dst[r] = g.ep.alpha * acc[i][j][r];
        }
      }
    }
    return;
  }
  const bf16_t* ybase = g.ep.mask_y ? g.ep.mask_y + czoff : nullptr;
  float cs[TN][4];  // column sums of the stored values (bias gradient of the masked layer)
#pragma unroll
  for (int j = 0; j < TN; ++j) cs[j][0] = cs[j][1] = cs[j][2] = cs[j][3] = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WM + i * 16 + (lane & 15);
    if (m >= g.M) continue;
    const long roff = split_off(g.C.sp, m);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
      if (n >= g.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = g.ep.alpha * acc[i][j][r];
      const bool full = cvec && n + 3 < g.N;
      if (ybase) {  // backward of ReLU + dropout from the layer's output (launcher: N % 4 == 0, aligned rows)
        const uint2 y4 = *reinterpret_cast<const uint2*>(ybase + roff + n);
        const unsigned yw[4] = {y4.x & 0xffffu, y4.x >> 16, y4.y & 0xffffu, y4.y >> 16};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = (yw[r] & 0x7fffu) != 0 ? v[r] * g.ep.mask_scale : 0.f;  // +-0 -> no gradient
          cs[j][r] += v[r];
        }
      }
      if (g.ep.bias && lead) {
        const float* bias = g.ep.bias + zq * g.ep.bias_zo;
        if (full) {
          const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += bias[n + r];
        }
      }
      if (g.ep.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (g.ep.act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(v[r]);  // (as the fast forms: a result must not depend on the form)
      }
      if (g.ep.drop_p > 0.f) {
        const uint64_t e0 = ((uint64_t)zb * g.M + m) * (uint64_t)g.N + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= drop_scale(g.ep.seed, e0 + r, g.ep.drop_p, inv_keep);
      }
      if (full) {
        if (rbase && lead) {
          const float4 r4 = *reinterpret_cast<const float4*>(rbase + roff + n);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        if (g.splitk > 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(cbase + roff + n + r, v[r]);
          continue;
        }
        if (g.ep.accumulate) {
          const float4 c4 = *reinterpret_cast<const float4*>(cbase + roff + n);
          v[0] += c4.x; v[1] += c4.y; v[2] += c4.z; v[3] += c4.w;
        }
        if (cbase) *reinterpret_cast<float4*>(cbase + roff + n) = make_float4(v[0], v[1], v[2], v[3]);
        if (hbase) *reinterpret_cast<uint2*>(hbase + roff + n) = pack_bf16x4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= g.N) continue;
          float x = v[r];
          if (rbase && lead) x += rbase[roff + n + r];
          if (g.splitk > 1) { atomicAdd(cbase + roff + n + r, x); continue; }
          if (g.ep.accumulate) x += cbase[roff + n + r];
          if (cbase) cbase[roff + n + r] = x;
          if (hbase) hbase[roff + n + r] = (bf16_t)(pack_bf16x4(x, 0.f, 0.f, 0.f).x & 0xffffu);
        }
      }
    }
  }
  if ((g.ep.colsum || g.ep.colsum_part) && ybase) {
    float* prow = g.ep.colsum_part ? g.ep.colsum_part + ((long)(m0 / BM) * 2 + wm) * g.N : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = row16_sum(cs[j][r]);
        if ((lane & 15) == 0 && n + r < g.N) {
          if (prow) prow[n + r] = t;
          else atomicAdd(g.ep.colsum + n + r, t);
        }
      }
    }
  }
}

#define S2ST_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x0f70)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// (the nt cache policy on these loads was measured: +0.45 ... +1.3 ms per step, profiles/r06_dma_nontemporal_ab.txt -- every panel
//  is re-read by the tiles beside it)
template <bool KM, int ROWS, int NW>
struct Dma {
  static constexpr int NI = ROWS / (8 * NW);  // wave-instructions per wave per stage (1 KiB each)
  static constexpr int PITCH = KM ? 128 : ROWS * 2;
  static constexpr int BYTES = ROWS * 128;
  static constexpr int KR_PER_I = 1024 / PITCH;  // rows-contiguous: k rows per wave-instruction
  const bf16_t* p[NI];  // KM: this lane's row (clamped) + chunk offset ; !KM: base + clamped first row of its chunk
  long ld;              // !KM: element stride of one k row
  int kr0[NI];          // !KM: k row (within the stage) this lane fetches in instruction j
  int chunk8;           // KM: 8 * (source chunk of this lane)

  __device__ __forceinline__ void init(const GemmOperand& X, const bf16_t* base, int r0, int R, int wave, int lane) {
    if (KM) {
      chunk8 = 8 * ((lane & 7) ^ (lane >> 3));
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = r0 + wave * (ROWS / NW) + j * 8 + (lane >> 3);
        p[j] = base + split_off(X.sp, min(row, R - 1));
      }
    } else {
      ld = X.sp.ld;
      constexpr int LPR = PITCH / 16;  // lanes (16-byte slots) per k row
      const int slot = lane % LPR;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int kr = wave * (64 / NW) + j * KR_PER_I + lane / LPR;
        kr0[j] = kr;
        const int ch = ((((slot >> 1) ^ kperm<ROWS>(kr)) << 1) | (slot & 1));
        p[j] = base + min(r0 + ch * 8, ((R + 7) & ~7) - 8);
      }
    }
  }

  // issue the DMA of K-step starting at kt into the stage image at `img` (wave-uniform)
  __device__ __forceinline__ void issue(unsigned char* img, int kt, int K, int wave) const {
    if (KM) {
      const int ko = min(kt + chunk8, ((K + 7) & ~7) - 8);  // stay inside the padded row
#pragma unroll
      for (int j = 0; j < NI; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(p[j] + ko), (lptr_t)(img + (wave * (ROWS / NW) + j * 8) * 128), 16, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const long k = min(kt + kr0[j], K - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(p[j] + k * ld),
                                         (lptr_t)(img + (wave * (64 / NW) + j * KR_PER_I) * PITCH), 16, 0, 0);
      }
    }
  }

  // the same, one wave-instruction (piece j of NI) at a time: for loops that place the pieces between MFMAs themselves
  __device__ __forceinline__ void issue_piece(unsigned char* img, int kt, int K, int wave, int j) const {
    if (KM) {
      const int ko = min(kt + chunk8, ((K + 7) & ~7) - 8);
      __builtin_amdgcn_global_load_lds((gptr_t)(p[j] + ko), (lptr_t)(img + (wave * (ROWS / NW) + j * 8) * 128), 16, 0, 0);
    } else {
      const long k = min(kt + kr0[j], K - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(p[j] + k * ld), (lptr_t)(img + (wave * (64 / NW) + j * KR_PER_I) * PITCH), 16, 0, 0);
    }
  }

  // zero k >= kv (valid k of this stage) in the landed image
  __device__ static __forceinline__ void sanitize(unsigned char* img, int kv, int tid) {
    for (int f = tid; f < ROWS * 8; f += 64 * NW) {
      if (KM) {
        const int row = f >> 3, slot = f & 7;
        const int nv = kv - 8 * (slot ^ (row & 7));
        if (nv < 8) {
          uint4* q = reinterpret_cast<uint4*>(img + row * 128 + slot * 16);
          *q = mask_tail(*q, nv > 0 ? nv : 0);
        }
      } else {
        constexpr int RC = ROWS / 8;
        if (f / RC >= kv) *reinterpret_cast<uint4*>(img + (long)f * 16) = make_uint4(0, 0, 0, 0);
      }
    }
  }
};

// scheduling-order helpers (the builtin wants literal counts): TM_ x (TN_ MFMAs, then this row's share of TOTAL
// instructions of class MASK2)
template <int MASK, int N>
__device__ __forceinline__ void sched_group() {
  if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
}
template <int I, int TM_, int TN_, int TOTAL, int SG2>
__device__ __forceinline__ void sched_rows() {
  if constexpr (I < TM_) {
    sched_group<0x008, TN_>();
    sched_group<SG2, (TOTAL * (I + 1)) / TM_ - (TOTAL * I) / TM_>();
    sched_rows<I + 1, TM_, TN_, TOTAL, SG2>();
  }
}

}  // namespace
