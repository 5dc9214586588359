// Fused multi-head attention for gfx950 (fast / bf16-operand mode): scores, key-padding + causal masks,
// fp32 online softmax, dropout on the probabilities and the P*V product in ONE kernel -- the
// [B,H,T,S] score / probability tensors of the unfused path are never written to HBM -- plus the two
// backward kernels (dK,dV and dQ) that recompute the probabilities from the saved log-sum-exp.
//
// Reference arithmetic replaced: fairseq/modules/multihead_attention.py:224 (q * dh^-0.5), :332 (bmm),
// :343-355 (masks -> -inf), :360-366 (fp32 softmax, dropout), :367 (bmm) and the same steps inside
// F.multi_head_attention_forward (:170-192); backward = the autograd of those ops.
//
// Layout / MFMA plan (v_mfma_f32_16x16x32_bf16, accumulator: col = lane & 15, row = 4 (lane >> 4) + r):
//  * a wave owns 16 "column" items (queries in fwd / dQ, keys in dK,dV); their bf16 fragments stay in
//    registers for the whole kernel; the other side is streamed through LDS in tiles of 32 rows;
//  * the score tile is computed TRANSPOSED (rows = streamed items on the accumulator rows), so the
//    probabilities a lane holds are exactly the B-operand fragment of the next MFMA (which sums over
//    the streamed index): no LDS round trip, no cross-lane movement.  k-slot (g, j) of that MFMA is
//    streamed row 4g + j (j < 4) / 16 + 4g + (j - 4); the A operand (V^T, dO^T, Q^T or K^T) is read
//    from a natural [row][dh] LDS image with ds_read_b64_tr_b16 in the same order;
//  * LDS images: "row image" [32][dh] with 16-byte chunk c at c ^ (row & (CH-1)) for ds_read_b128
//    fragments, "tr image" [32][dh] with 32-byte pair c at c ^ f(row) for the transposed reads.
// Softmax state (running max, partial sums) lives per lane: the 4 lanes that share a column reduce the
// max with two shuffles per tile and the sum once at the end.
#include <cstdlib>
#include <type_traits>

#include "s2st_ops.h"
#include "s2st_prof.h"

namespace {

typedef unsigned short bf16_t;

template <int DH>
struct Img {
  static constexpr int PITCH = DH * 2;     // bytes per row
  static constexpr int CH = DH / 8;        // 16-byte chunks per row
  static constexpr int BYTES = 32 * PITCH;
  __device__ static __forceinline__ int trkey(int row) { return DH >= 128 ? (row & 7) : ((row >> 1) & 3); }
  // store 16-byte chunk `ch` of row `row` into both images' positions
  __device__ static __forceinline__ int row_off(int row, int ch) { return row * PITCH + ((ch ^ (row & (CH - 1))) << 4); }
  __device__ static __forceinline__ int tr_off(int row, int ch) {
    return row * PITCH + (((((ch >> 1) ^ trkey(row)) << 1) | (ch & 1)) << 4);
  }
  // A-style fragment (rows rt..rt+15, k = 32 ks + 8 (l>>4) + j) from a row image
  __device__ static __forceinline__ bf16x8 row_frag(const unsigned char* img, int rt, int ks, int lane) {
    const int row = rt + (lane & 15), ch = 4 * ks + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(img + row_off(row, ch));
  }
  // transposed fragment from a tr image: lane -> column dt*16 + (l & 15); k-slot (g, j) = row
  // 4g + j (j < 4) / 16 + 4g + (j - 4)
  __device__ static __forceinline__ bf16x8 tr_frag(const unsigned char* img, int dt, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r0 = 4 * g + q, r1 = 16 + 4 * g + q;
    s16x4 lo = lds_read_tr16(img + r0 * PITCH + ((dt ^ trkey(r0)) << 5) + 8 * p);
    s16x4 hi = lds_read_tr16(img + r1 * PITCH + ((dt ^ trkey(r1)) << 5) + 8 * p);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
};

// cooperative staging of a [32][DH] bf16 tile (rows r0.., zero beyond nrows): fetch() issues the
// 16-byte global loads into registers (so they fly during the MFMAs of the previous tile), commit()
// writes them into a row image and/or a tr image.  256 threads; DH/8 chunks per row.
template <int DH, int NT>
struct TileRegs {
  static constexpr int CH = DH / 8, N = 32 * CH / NT;
  uint4 r[N];
  __device__ __forceinline__ void fetch(const bf16_t* __restrict__ base, long ld, int r0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int f = tid + NT * i, row = f / CH, ch = f - row * CH;
      // (rows past the end: the load goes to the last row and is replaced by zeros -- a select, not a branch)
      const uint4 v = *reinterpret_cast<const uint4*>(base + (long)min(r0 + row, nrows - 1) * ld + ch * 8);
      r[i] = (r0 + row < nrows) ? v : make_uint4(0, 0, 0, 0);
    }
  }
  __device__ __forceinline__ void commit(unsigned char* rimg, unsigned char* timg, int tid) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int f = tid + NT * i, row = f / CH, ch = f - row * CH;
      if (rimg) *reinterpret_cast<uint4*>(rimg + Img<DH>::row_off(row, ch)) = r[i];
      if (timg) *reinterpret_cast<uint4*>(timg + Img<DH>::tr_off(row, ch)) = r[i];
    }
  }
};

// sum over the 8 bf16 pairs of two 16-byte chunks (fp32 products and adds)
__device__ __forceinline__ float dot_bf16x8(const uint4& a, const uint4& b) {
  const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s += __uint_as_float(aw[i] << 16) * __uint_as_float(bw[i] << 16);
    s += __uint_as_float(aw[i] & 0xffff0000u) * __uint_as_float(bw[i] & 0xffff0000u);
  }
  return s;
}

// out[row] = sum over the row of x * y for the 32 rows two staged tiles hold (the DH / 8 chunks of a row sit in
// DH / 8 neighbouring lanes: a DPP row of 16 for DH = 128, half of one for DH = 64)
template <int DH, int NT>
__device__ __forceinline__ void tile_rowdots(const TileRegs<DH, NT>& x, const TileRegs<DH, NT>& y, float* out, int tid) {
  constexpr int CH = DH / 8;
#pragma unroll
  for (int i = 0; i < TileRegs<DH, NT>::N; ++i) {
    const int f = tid + NT * i, row = f / CH, ch = f - row * CH;
    float s = dot_bf16x8(x.r[i], y.r[i]);
    if (CH == 16) s = row16_sum(s);
    else {
#pragma unroll
      for (int m = CH / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    }
    if (ch == 0) out[row] = s;
  }
}

// register fragments of 16 rows (clamped) of a [rows][DH] bf16 matrix, A-style (== B-style of X^T)
template <int DH>
__device__ __forceinline__ void load_frags(const bf16_t* __restrict__ base, long ld, int r0, int nrows, int lane,
                                           bf16x8 (&fr)[DH / 32]) {
  const int row = min(r0 + (lane & 15), nrows - 1);
#pragma unroll
  for (int ks = 0; ks < DH / 32; ++ks)
    fr[ks] = *reinterpret_cast<const bf16x8*>(base + (long)row * ld + 32 * ks + 8 * (lane >> 4));
}

__device__ __forceinline__ unsigned pack2(float a, float b) { return pack_bf16x4(a, b, 0.f, 0.f).x; }


__device__ __forceinline__ bf16x8 pack_frag(const float (&v)[8]) {
  union { uint4 u; bf16x8 f; } cv;
  const uint2 a = pack_bf16x4(v[0], v[1], v[2], v[3]), b = pack_bf16x4(v[4], v[5], v[6], v[7]);
  cv.u = make_uint4(a.x, a.y, b.x, b.y);
  return cv.f;
}

// XCD-aware workgroup order: hardware deals consecutive workgroups (x fastest) to the 8 XCDs in turn, so the blocks of one
// (batch, head) pair -- which all read the same Q / K / V / dO tiles -- used to land on 8 different L2s (PMC: the backward
// fetched 68 MB per launch for 17 MB of operands).  The linear id is re-read as (XCD = id % 8, slot = id / 8): XCD x
// owns pair 8 s + x of every group s of 8 pairs, for all of that pair's blocks.  Pairs beyond the last full group keep
// the plain order.  (`plain` = 1 restores launch order: round 3's A/B form, no longer switchable.)
__device__ __forceinline__ void attn_block(int plain, int& bx, int& bh) {
  const int nbx = gridDim.x, nbh = gridDim.y;
  bx = blockIdx.x;
  bh = blockIdx.y;
  const int L = bh * nbx + bx, full = (nbh >> 3) << 3;
  if (!plain && L < full * nbx) {
    const int x = L & 7, j = L >> 3, s = j / nbx;
    bh = 8 * s + x;
    bx = j - s * nbx;
  }
}

struct AttnArgs {
  const bf16_t *q, *k, *v;       // bf16 projections; row (b, t) at base + (b * rows + t) * ld + h * DH
  long ldq, ldk, ldv;
  float* o;                       // fwd out: [B*T][H*DH] fp32 (ld = H*DH)
  bf16_t* oh;                     // bf16 copy of o (or null)
  float* lse;                     // [B*H*T]
  const int* klen;                // [B] valid keys (or null)
  int B, H, T, S, causal;
  float scale, drop_p;
  uint64_t seed;
  int ld_drop;                    // row stride of the dropout index space (== unfused path's score ld)
  // backward
  const bf16_t* doh;              // bf16 dO [B*T][H*DH]
  const float* dvec;              // D[b,h,t] = rowsum(dO * O)
  int plain_order;                // 1: blocks in launch order (see attn_block)
#ifdef S2ST_ATTN_STAMP  // tools/attn_stamp.sh: a PRIVATE build that writes per-workgroup clock stamps (8 longs each)
  long* stamp;
#define ATTN_STAMP(k) do { if (threadIdx.x == 0) a.stamp[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = clock64(); } while (0)
#else
#define ATTN_STAMP(k) do { } while (0)
#endif
  float *dq, *dk, *dv;            // fp32 gradients, addressed like q / k / v (ldq / ldk / ldv); may be null
  bf16_t *dqh, *dkh, *dvh;        // optional bf16 gradients (the operand of the projections' backward GEMMs)
  float *dbq, *dbk, *dbv;         // optional: projection bias gradients += column sums (head h at + h * DH)
  // db_part != 0: dbq / dbk / dbv point at PARTIAL-sum arrays [slot][H * DH] instead (slot = (b * blocks + block) * waves
  // + wave: every element written exactly once, plain stores); the launcher folds them in slot order (no atomics)
  int db_part;
};

// store a [d][col] accumulator tile set as row `col`: fp32 and/or bf16, and add its column sums (over the
// wave's 16 rows, rows >= limit excluded by `on`) to a bias gradient
template <int DT>
__device__ __forceinline__ void store_grad(const f32x4 (&t)[DT], float scale, bool on, float* fp, bf16_t* hp,
                                           float* db, int lane, bool db_store = false) {
  const int g = lane >> 4;
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = on ? t[d][r] * scale : 0.f;
    if (on) {
      if (fp) *reinterpret_cast<float4*>(fp + 16 * d + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
      if (hp) *reinterpret_cast<uint2*>(hp + 16 * d + 4 * g) = pack_bf16x4(v[0], v[1], v[2], v[3]);
    }
    if (db) {
      // column sums over the wave's 16 rows (the 16 lanes of a DPP row): four DPP adds per value (s2st_asm.h) -- round 2
      // used four ds_bpermute shuffles per value here (128 per accumulator set on the data path: +3.8 % of a step)
      float c[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = row16_sum(v[r]);
      if ((lane & 15) == 0) {
        if (db_store) *reinterpret_cast<float4*>(db + 16 * d + 4 * g) = make_float4(c[0], c[1], c[2], c[3]);
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(db + 16 * d + 4 * g + r, c[r]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward: grid (ceil(T / 64), B * H); 4 waves x 16 queries; key tiles of 32
// ------------------------------------------------------------------------------------------------
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 3) void flash_fwd_kernel(AttnArgs a) {
  constexpr int KS = DH / 32, DT = DH / 16;
  if (a.T <= 0 || a.S <= 0) return;  // empty problem (kernel preload)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * Img<DH>::BYTES];
  unsigned char* kimg = smem;                      // row image of the K tile
  unsigned char* vimg = smem + Img<DH>::BYTES;     // tr image of the V tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  int bxr, bh;
  attn_block(a.plain_order, bxr, bh);
  const int b = bh / a.H, h = bh - b * a.H;
  const int q0 = bxr * (16 * NW) + wave * 16;
  const int qi = q0 + (lane & 15);                 // this lane's query (column)
  const bf16_t* qb = a.q + (long)b * a.T * a.ldq + h * DH;
  const bf16_t* kb = a.k + (long)b * a.S * a.ldk + h * DH;
  const bf16_t* vb = a.v + (long)b * a.S * a.ldv + h * DH;
  int klim = a.klen ? min((int)a.klen[b], a.S) : a.S;
  const int kmax = a.causal ? min(klim, (bxr + 1) * (16 * NW)) : klim;  // keys any query of the block sees

  bf16x8 qf[KS];
  load_frags<DH>(qb, a.ldq, q0, a.T, lane, qf);
  f32x4 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  const uint64_t drow = ((uint64_t)bh * a.T + qi) * (uint64_t)a.ld_drop;

  // two register sets of staged tiles (even / odd): a tile's loads are issued two tiles ahead (see the backward)
  struct Stage { TileRegs<DH, 64 * NW> k, v; } stg[2];
  auto fetch = [&](Stage& st, int kt) {
    st.k.fetch(kb, a.ldk, kt, a.S, tid);
    st.v.fetch(vb, a.ldv, kt, a.S, tid);
  };
  if (kmax > 0) fetch(stg[0], 0);
  if (kmax > 32) fetch(stg[1], 32);
  auto tile = [&](Stage& st, const int kt) {
    __syncthreads();  // everyone is done with the previous tile's images
    st.k.commit(kimg, nullptr, tid);
    st.v.commit(nullptr, vimg, tid);
    __syncthreads();
    if (kt + 64 < kmax) fetch(st, kt + 64);  // flies during this tile's and the next tile's MFMAs
    // S^T tiles: rows = keys kt + 16 t2 + 4g + r, col = query
    f32x4 x[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      x[t2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        x[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Img<DH>::row_frag(kimg, 16 * t2, ks, lane), qf[ks], x[t2], 0, 0, 0);
    }
    float s[8];
    float mt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = kt + 16 * (e >> 2) + 4 * g + (e & 3);
      const bool ok = key < klim && (!a.causal || key <= qi);
      s[e] = ok ? x[e >> 2][e & 3] * a.scale : -INFINITY;
      mt = fmaxf(mt, s[e]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 16));
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float mn = fmaxf(m, mt);
    const float mu = mn == -INFINITY ? 0.f : mn;
    const float alpha = __expf(m - mu);  // m = -inf -> 0
    float p[8], ps = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p[e] = __expf(s[e] - mu);
      ps += p[e];
    }
    if (a.drop_p > 0.f) {  // (one wave-uniform branch per tile, not one per element)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        p[e] *= drop_scale(a.seed, drow + (uint64_t)(kt + 16 * (e >> 2) + 4 * g + (e & 3)), a.drop_p, inv_keep);
    }
    l = l * alpha + ps;
    m = mn;
    const bf16x8 pf = pack_frag(p);
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      o[d] *= alpha;
      o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Img<DH>::tr_frag(vimg, d, lane), pf, o[d], 0, 0, 0);
    }
  };
  for (int kt = 0; kt < kmax; kt += 64) {
    tile(stg[0], kt);
    if (kt + 32 < kmax) tile(stg[1], kt + 32);
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float inv_l = l > 0.f ? 1.f / l : 0.f;
  if (qi < a.T) {
    const long ro = ((long)b * a.T + qi) * ((long)a.H * DH) + h * DH;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const float4 v = make_float4(o[d][0] * inv_l, o[d][1] * inv_l, o[d][2] * inv_l, o[d][3] * inv_l);
      *reinterpret_cast<float4*>(a.o + ro + 16 * d + 4 * g) = v;
      if (a.oh) *reinterpret_cast<uint2*>(a.oh + ro + 16 * d + 4 * g) = pack_bf16x4(v.x, v.y, v.z, v.w);
    }
    if (g == 0 && a.lse) a.lse[(long)bh * a.T + qi] = l > 0.f ? m + __logf(l) : -INFINITY;
  }
}

// D[b,h,t] = sum_d dO[b,t,h,d] * O[b,t,h,d]: DH/4 lanes per (b,t,h) row, one float4 pair per lane
template <int DH>
__global__ __launch_bounds__(256) void attn_dvec_kernel(const float* __restrict__ dO, const float* __restrict__ O,
                                                        float* __restrict__ D, int B, int H, int T) {
  constexpr int LPR = DH / 4;  // lanes per row (32 or 16)
  const long row = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;  // over B*T*H
  const int l = threadIdx.x % LPR;
  const bool ok = row < (long)B * T * H;
  float s = 0.f;
  int h = 0;
  long bt = 0;
  if (ok) {
    h = (int)(row % H);
    bt = row / H;
    const long o = bt * ((long)H * DH) + (long)h * DH + 4 * l;
    const float4 a = *reinterpret_cast<const float4*>(dO + o);
    const float4 c = *reinterpret_cast<const float4*>(O + o);
    s = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
  }
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (ok && l == 0) {
    const int t = (int)(bt % T), bb = (int)(bt / T);
    D[((long)bb * H + h) * T + t] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// backward dK, dV: grid (ceil(S / 64), B * H); 4 waves x 16 keys (columns); query tiles of 32 (rows)
//   X[q][key] = Q K^T ; P = exp(scale X - lse[q]) ; Pd = dropout(P)
//   dV^T[d][key] += dO^T[d][q] Pd[q][key]
//   dP[q][key] = dO[q][:] . V[key][:] ; dS = P * (mask * dP / keep - D[q])
//   dK^T[d][key] += Q^T[d][q] dS[q][key]   (scaled at the end)
// ------------------------------------------------------------------------------------------------
template <int DH, int NW>
__device__ __forceinline__ void flash_bwd_kv_body(const AttnArgs& a, unsigned char* smem, const int bx, const int bh) {
  constexpr int KS = DH / 32, DT = DH / 16;
  unsigned char* q_row = smem;
  unsigned char* q_tr = smem + Img<DH>::BYTES;
  unsigned char* do_row = smem + 2 * Img<DH>::BYTES;
  unsigned char* do_tr = smem + 3 * Img<DH>::BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int b = bh / a.H, h = bh - b * a.H;
  const int k0 = bx * (16 * NW) + wave * 16;
  const int ki = k0 + (lane & 15);  // this lane's key (column)
  const bf16_t* qb = a.q + (long)b * a.T * a.ldq + h * DH;
  const bf16_t* kb = a.k + (long)b * a.S * a.ldk + h * DH;
  const bf16_t* vb = a.v + (long)b * a.S * a.ldv + h * DH;
  const bf16_t* dob = a.doh + (long)b * a.T * ((long)a.H * DH) + h * DH;
  const int klim = a.klen ? min((int)a.klen[b], a.S) : a.S;
  const bool key_ok = ki < klim;
  const int qbeg = a.causal ? (bx * (16 * NW)) & ~31 : 0;  // queries < first key of the block see none of it

  ATTN_STAMP(0);
  bf16x8 kf[KS], vf[KS];
  load_frags<DH>(kb, a.ldk, k0, a.S, lane, kf);
  load_frags<DH>(vb, a.ldv, k0, a.S, lane, vf);
  f32x4 dk[DT], dv[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) dk[d] = dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;

  // D[q] = rowsum(dO * O) of the tile's 32 queries (a.dvec == null: no separate kernel ran): the O tile travels in
  // registers beside the dO tile (same thread -> chunk map: the lanes that hold a row's chunks are neighbours), the
  // row sums land in LDS with the tile images
  __shared__ float dsh[32];
  // the tile's 32 log-sum-exp values travel the same way (one float per thread 0 .. 31, fetched a tile ahead): no
  // scattered global loads between the score MFMAs and the exponentials
  __shared__ float lsh[32];
  const float* lseb = a.lse + (long)bh * a.T;
  float lse_nx = 0.f;
  const bool own_d = a.dvec == nullptr;
  const bf16_t* ob = own_d ? a.oh + (long)b * a.T * ((long)a.H * DH) + h * DH : nullptr;
  // One register set of staged tiles: the next tile's global loads are issued right after this tile went to LDS.  (Two
  // sets, loads two tiles ahead, were measured: -0.7 us per launch at +25 VGPRs -- with the straight-line element code
  // below that spills past the 256 registers two waves per SIMD leave; the forward kernel keeps two sets.)
  struct Stage {
    TileRegs<DH, 64 * NW> q, d, o;
    float lse;
  } stg[1];
  auto fetch = [&](Stage& st, int qt) {
    st.q.fetch(qb, a.ldq, qt, a.T, tid);
    st.d.fetch(dob, (long)a.H * DH, qt, a.T, tid);
    if (own_d) st.o.fetch(ob, (long)a.H * DH, qt, a.T, tid);
    if (tid < 32) st.lse = lseb[min(qt + tid, a.T - 1)];
  };
  if (qbeg < a.T) fetch(stg[0], qbeg);
  auto tile = [&](Stage& st, const int qt) {
    __syncthreads();
    st.q.commit(q_row, q_tr, tid);
    st.d.commit(do_row, do_tr, tid);
    if (own_d) tile_rowdots<DH, 64 * NW>(st.d, st.o, dsh, tid);
    if (tid < 32) lsh[tid] = st.lse;
    __syncthreads();
    if (qt == qbeg) ATTN_STAMP(1);        // first tile in LDS
    if (qt == qbeg + 32) ATTN_STAMP(2);   // second tile in LDS (= one full iteration later)
    if (qt + 32 < a.T) fetch(st, qt + 32);
    f32x4 x[2], dp[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      x[t2] = dp[t2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        x[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Img<DH>::row_frag(q_row, 16 * t2, ks, lane), kf[ks], x[t2], 0, 0, 0);
        dp[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Img<DH>::row_frag(do_row, 16 * t2, ks, lane), vf[ks], dp[t2], 0, 0, 0);
      }
    }
    // Element work as straight-line code (selects, no per-element branches: the first form's `if (ok)` / `if (dropout)`
    // per element were 16 exec-mask branches per tile): the two wave-uniform choices -- dropout on / off, D from LDS or
    // from the row kernel's vector -- are made once per tile; masked elements compute on clamped indices and are zeroed.
    float pd[8], ds[8], keep[8], dvl[8];
    const uint64_t d0 = ((uint64_t)bh * a.T + qt + 4 * g) * (uint64_t)a.ld_drop + ki;  // dropout index of element 0
    if (a.drop_p > 0.f) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        keep[e] = drop_scale(a.seed, d0 + (uint64_t)((16 * (e >> 2) + (e & 3)) * a.ld_drop), a.drop_p, inv_keep);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) keep[e] = 1.f;
    }
    if (own_d) {
#pragma unroll
      for (int e = 0; e < 8; ++e) dvl[e] = dsh[16 * (e >> 2) + 4 * g + (e & 3)];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dvl[e] = a.dvec[(long)bh * a.T + min(qt + 16 * (e >> 2) + 4 * g + (e & 3), a.T - 1)];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int qq = 16 * (e >> 2) + 4 * g + (e & 3), q = qt + qq;
      const bool ok = key_ok && q < a.T && (!a.causal || ki <= q);
      const float p = ok ? __expf(x[e >> 2][e & 3] * a.scale - lsh[qq]) : 0.f;
      pd[e] = p * keep[e];
      ds[e] = p * (keep[e] * dp[e >> 2][e & 3] - dvl[e]);
    }
    if (qt == qbeg + 32) ATTN_STAMP(3);   // scores, dP, exponentials, dropout of the second tile done
    const bf16x8 pf = pack_frag(pd), sf = pack_frag(ds);
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Img<DH>::tr_frag(do_tr, d, lane), pf, dv[d], 0, 0, 0);
      dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Img<DH>::tr_frag(q_tr, d, lane), sf, dk[d], 0, 0, 0);
    }
  };
  for (int qt = qbeg; qt < a.T; qt += 32) tile(stg[0], qt);
  ATTN_STAMP(4);  // loop done
  {
    const bool on = ki < a.S;
    const long ko = ((long)b * a.S + min(ki, a.S - 1)) * a.ldk + h * DH;
    const long vo = ((long)b * a.S + min(ki, a.S - 1)) * a.ldv + h * DH;
    // (partial-sum mode: this wave's slot of the [slot][H * DH] arrays)
    const long slot = a.db_part ? ((long)(b * (int)((a.S + 16 * NW - 1) / (16 * NW)) + bx) * NW + wave) * ((long)a.H * DH) : 0;
    store_grad<DT>(dk, a.scale, on, a.dk ? a.dk + ko : nullptr, a.dkh ? a.dkh + ko : nullptr,
                   a.dbk ? a.dbk + slot + h * DH : nullptr, lane, a.db_part != 0);
    store_grad<DT>(dv, 1.f, on, a.dv ? a.dv + vo : nullptr, a.dvh ? a.dvh + vo : nullptr,
                   a.dbv ? a.dbv + slot + h * DH : nullptr, lane, a.db_part != 0);
  }
  ATTN_STAMP(5);  // gradients stored
}

// ------------------------------------------------------------------------------------------------
// backward dQ: grid (ceil(T / 64), B * H); 4 waves x 16 queries (columns); key tiles of 32 (rows)
//   X^T[key][q] = K Q^T ; dP^T[key][q] = V dO^T ; dS^T = P^T * (mask dP^T / keep - D[q])
//   dQ^T[d][q] += K^T[d][key] dS^T[key][q]
// ------------------------------------------------------------------------------------------------
template <int DH, int NW>
__device__ __forceinline__ void flash_bwd_q_body(const AttnArgs& a, unsigned char* smem, const int bx, const int bh) {
  constexpr int KS = DH / 32, DT = DH / 16;
  unsigned char* k_row = smem;
  unsigned char* k_tr = smem + Img<DH>::BYTES;
  unsigned char* v_row = smem + 2 * Img<DH>::BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int b = bh / a.H, h = bh - b * a.H;
  const int q0 = bx * (16 * NW) + wave * 16;
  const int qi = q0 + (lane & 15);
  const bf16_t* qb = a.q + (long)b * a.T * a.ldq + h * DH;
  const bf16_t* kb = a.k + (long)b * a.S * a.ldk + h * DH;
  const bf16_t* vb = a.v + (long)b * a.S * a.ldv + h * DH;
  const bf16_t* dob = a.doh + (long)b * a.T * ((long)a.H * DH) + h * DH;
  const int klim = a.klen ? min((int)a.klen[b], a.S) : a.S;
  const int kmax = a.causal ? min(klim, (bx + 1) * (16 * NW)) : klim;

  bf16x8 qf[KS], dof[KS];
  load_frags<DH>(qb, a.ldq, q0, a.T, lane, qf);
  load_frags<DH>(dob, (long)a.H * DH, q0, a.T, lane, dof);
  f32x4 dq[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  const bool q_ok = qi < a.T;
  const long r = (long)bh * a.T + min(qi, a.T - 1);
  const float lse = a.lse[r];
  float dvec;
  if (a.dvec) dvec = a.dvec[r];
  else {  // D[q] from the bf16 dO fragments this lane already holds and the matching fragments of O
    bf16x8 of[KS];
    load_frags<DH>(a.oh + (long)b * a.T * ((long)a.H * DH) + h * DH, (long)a.H * DH, q0, a.T, lane, of);
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      union { bf16x8 f; uint4 u; } cx, cy;
      cx.f = dof[ks];
      cy.f = of[ks];
      s += dot_bf16x8(cx.u, cy.u);
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    dvec = s;
  }
  const uint64_t drow = (uint64_t)r * a.ld_drop;

  struct Stage { TileRegs<DH, 64 * NW> k, v; } stg[1];  // (one register set: see the dK/dV body)
  auto fetch = [&](Stage& st, int kt) {
    st.k.fetch(kb, a.ldk, kt, a.S, tid);
    st.v.fetch(vb, a.ldv, kt, a.S, tid);
  };
  if (kmax > 0) fetch(stg[0], 0);
  auto tile = [&](Stage& st, const int kt) {
    __syncthreads();
    st.k.commit(k_row, k_tr, tid);
    st.v.commit(v_row, nullptr, tid);
    __syncthreads();
    if (kt + 32 < kmax) fetch(st, kt + 32);
    f32x4 x[2], dp[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      x[t2] = dp[t2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        x[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Img<DH>::row_frag(k_row, 16 * t2, ks, lane), qf[ks], x[t2], 0, 0, 0);
        dp[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Img<DH>::row_frag(v_row, 16 * t2, ks, lane), dof[ks], dp[t2], 0, 0, 0);
      }
    }
    float ds[8], keep[8];  // (straight-line element work: see the dK/dV body)
    if (a.drop_p > 0.f) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        keep[e] = drop_scale(a.seed, drow + (uint64_t)(kt + 16 * (e >> 2) + 4 * g + (e & 3)), a.drop_p, inv_keep);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) keep[e] = 1.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = kt + 16 * (e >> 2) + 4 * g + (e & 3);
      const bool ok = q_ok && key < klim && (!a.causal || key <= qi);
      const float p = ok ? __expf(x[e >> 2][e & 3] * a.scale - lse) : 0.f;
      ds[e] = p * (keep[e] * dp[e >> 2][e & 3] - dvec);
    }
    const bf16x8 sf = pack_frag(ds);
#pragma unroll
    for (int d = 0; d < DT; ++d)
      dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Img<DH>::tr_frag(k_tr, d, lane), sf, dq[d], 0, 0, 0);
  };
  for (int kt = 0; kt < kmax; kt += 32) tile(stg[0], kt);
  {
    const long qo = ((long)b * a.T + min(qi, a.T - 1)) * a.ldq + h * DH;
    const long slot = a.db_part ? ((long)(b * (int)((a.T + 16 * NW - 1) / (16 * NW)) + bx) * NW + wave) * ((long)a.H * DH) : 0;
    store_grad<DT>(dq, a.scale, q_ok, a.dq ? a.dq + qo : nullptr, a.dqh ? a.dqh + qo : nullptr,
                   a.dbq ? a.dbq + slot + h * DH : nullptr, lane, a.db_part != 0);
  }
}

template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void flash_bwd_kv_kernel(AttnArgs a) {
  if (a.T <= 0 || a.S <= 0) return;  // empty problem (kernel preload)
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * Img<DH>::BYTES];
  int bx, bh;
  attn_block(a.plain_order, bx, bh);
  flash_bwd_kv_body<DH, NW>(a, smem, bx, bh);
}

template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void flash_bwd_q_kernel(AttnArgs a) {
  if (a.T <= 0 || a.S <= 0) return;
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * Img<DH>::BYTES];
  int bx, bh;
  attn_block(a.plain_order, bx, bh);
  flash_bwd_q_body<DH, NW>(a, smem, bx, bh);
}

// Both passes in ONE launch: workgroups [0, nkx) of a (b, h) row are key blocks (dK, dV), the rest query blocks (dQ).
// The two passes share nothing but their inputs, so one dispatch (one kernel boundary on the data-path stream instead of
// two -- ~4 us each between dependent kernels, profiles/r02_b_timeline_summary_all_dispatches.txt) lets the short
// query blocks fill the tail of the key blocks.
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void flash_bwd_kernel(AttnArgs a, int nkx) {
  if (a.T <= 0 || a.S <= 0) return;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * Img<DH>::BYTES];
  int bx, bh;
  attn_block(a.plain_order, bx, bh);
  if (bx < nkx) flash_bwd_kv_body<DH, NW>(a, smem, bx, bh);
  else flash_bwd_q_body<DH, NW>(a, smem, bx - nkx, bh);
}

// ------------------------------------------------------------------------------------------------
// Short-sequence backward (round 6): T, S <= 128 -- 50 % / 80 % of the bench's encoder / decoder
// batches.  The two-pass kernels above cost ~30 us per launch whatever the length (profiles/r05_attn_by_geometry.txt):
// 1.2 - 2.9 rounds of workgroups whose life is their fixed part -- two dependent round trips before the first tile is
// in LDS, two barriers + a global -> LDS commit per 32-row tile -- and the scores are computed twice (7 products).
// Here ONE workgroup of 8 waves owns a (batch, head) pair:
//   phase 0  Q, dO, K go to LDS once (<= 128 rows each, 96 KB at DH 128); the wave's 16 keys' K / V fragments to registers;
//            D[q] = rowsum(dO * O) and lse[q] to LDS;                                                  ONE barrier
//   phase A  wave w owns keys 16 w ..: per 32-query tile  X = Q K^T, dP = dO V^T, P, dS (the element work, ONCE),
//            dV += dO^T Pd, dK += Q^T dS, and dS^T goes to LDS as bf16 [key][query] (8-byte stores)    no barriers
//   phase B  (ONE barrier) wave w owns queries 16 w ..: dQ^T += K^T dS^T over the key tiles, the dS^T fragments read
//            back with ds_read_b64_tr_b16 -- 5 products instead of 7, every operand fetched from HBM once.
// LDS images are "dual": one copy of a tile serves BOTH the row-fragment reads (ds_read_b128, rows on lanes) and the
// transposed reads -- the 32-byte pair is XORed with the row's low bits as in Img<>, and the 16-byte half inside the pair
// with bit 3 of the row, so that the 16 rows of a b128 phase hit 16 different 16-byte slots and the 8 rows of a
// transposed-read phase 8 different pairs.
// Masks, dropout decisions (same (seed, element) hash and index space), bias-gradient partial slots and scaling are the
// two-pass kernels'; results differ from theirs only by summation order and by dS passing through bf16 once (it already
// does: it is the MFMA operand of dK and dQ there too).
// ------------------------------------------------------------------------------------------------
template <int DH>
struct Img2 {
  static constexpr int PITCH = DH * 2;  // bytes per row
  static constexpr int CH = DH / 8;     // 16-byte chunks per row
  __device__ static __forceinline__ int key(int row) { return DH >= 128 ? (row & 7) : ((row >> 1) & 3); }
  __device__ static __forceinline__ int flip(int row) { return (row >> 3) & 1; }
  __device__ static __forceinline__ int off(int row, int ch) {
    return row * PITCH + (((((ch >> 1) ^ key(row)) << 1) | ((ch & 1) ^ flip(row))) << 4);
  }
  // A-style fragment: rows rt .. rt + 15 on the lanes, k = 32 ks + 8 (l >> 4) + j
  __device__ static __forceinline__ bf16x8 row_frag(const unsigned char* img, int rt, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + off(rt + (lane & 15), 4 * ks + (lane >> 4)));
  }
  // transposed fragment: lane -> column 16 dt + (l & 15); k-slot (g, j) = row rb + 4g + j (j < 4) / rb + 16 + 4g + (j - 4)
  __device__ static __forceinline__ bf16x8 tr_frag(const unsigned char* img, int rb, int dt, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r0 = rb + 4 * g + q, r1 = r0 + 16;
    s16x4 lo = lds_read_tr16(img + off(r0, 2 * dt + (p >> 1)) + 8 * (p & 1));
    s16x4 hi = lds_read_tr16(img + off(r1, 2 * dt + (p >> 1)) + 8 * (p & 1));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
};

constexpr int SHORT_MAX = 128;  // longest sequence of the short forms (8 waves x 16 columns)
// NW = 8 waves: T, S <= 128, one workgroup per CU; NW = 4: T, S <= 64 in half the LDS, two workgroups per CU (batches of many
// short utterances have more (batch, head) pairs than the chip has CUs).  Images are sized by the tile-rounded lengths.
inline int short_lds_bytes(int dh, int T, int S) {
  const int Tt = (T + 31) & ~31, St = (S + 31) & ~31;
  return (2 * Tt + St) * dh * 2 + St * 256 + 2 * SHORT_MAX * 4;
}

template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void flash_bwd_short_kernel(AttnArgs a) {
  constexpr int KS = DH / 32, DT = DH / 16, NT = 64 * NW, CH = DH / 8;
  constexpr int RMAX = 16 * NW;            // rows per image at most
  constexpr int NI = RMAX * CH / NT;       // 16-byte chunks per thread and image (4 at DH 128, 2 at DH 64)
  constexpr int RPI = NT / CH;             // rows per load iteration (32 / 64 at 8 waves)
  if (a.T <= 0 || a.S <= 0) return;        // empty problem (kernel preload)
  using I = Img2<DH>;
  using IS = Img2<128>;                    // the dS^T image: rows = keys, 128 query columns
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  const int Tt = (a.T + 31) & ~31, St = (a.S + 31) & ~31;
  unsigned char* qimg = smem;
  unsigned char* doimg = qimg + Tt * I::PITCH;
  unsigned char* kimg = doimg + Tt * I::PITCH;
  unsigned char* dst = kimg + St * I::PITCH;
  float* dsh = reinterpret_cast<float*>(dst + St * 256);
  float* lsh = dsh + SHORT_MAX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int bh = blockIdx.x;
  const int b = bh / a.H, h = bh - b * a.H;
  const long ldo = (long)a.H * DH;
  const bf16_t* qb = a.q + (long)b * a.T * a.ldq + h * DH;
  const bf16_t* kb = a.k + (long)b * a.S * a.ldk + h * DH;
  const bf16_t* vb = a.v + (long)b * a.S * a.ldv + h * DH;
  const bf16_t* dob = a.doh + (long)b * a.T * ldo + h * DH;
  const int klim = a.klen ? min((int)a.klen[b], a.S) : a.S;
  const bool own_d = a.dvec == nullptr;
  const bf16_t* ob = own_d ? a.oh + (long)b * a.T * ldo + h * DH : nullptr;

  // ---- phase 0: every global load of the workgroup is issued before anything waits ------------------------------
  ATTN_STAMP(0);
  uint4 rq[NI], rd[NI], ro[NI], rk[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int f = tid + NT * i, row = f / CH, ch = f - row * CH;
    rq[i] = rd[i] = ro[i] = rk[i] = make_uint4(0, 0, 0, 0);
    if (RPI * i < Tt) {  // (wave-uniform: an iteration covers whole 32-row tiles)
      const long r = min(row, a.T - 1);
      const uint4 vq = *reinterpret_cast<const uint4*>(qb + r * a.ldq + ch * 8);
      const uint4 vd = *reinterpret_cast<const uint4*>(dob + r * ldo + ch * 8);
      if (row < a.T) { rq[i] = vq; rd[i] = vd; }
      if (own_d) {
        const uint4 vo = *reinterpret_cast<const uint4*>(ob + r * ldo + ch * 8);
        if (row < a.T) ro[i] = vo;
      }
    }
    if (RPI * i < St) {
      const uint4 vk = *reinterpret_cast<const uint4*>(kb + (long)min(row, a.S - 1) * a.ldk + ch * 8);
      if (row < a.S) rk[i] = vk;
    }
  }
  const int k0 = wave * 16, ki = k0 + (lane & 15);  // phase A: this lane's key (column)
  const bool key_ok = ki < klim;
  const bool run_a = k0 < St;
  bf16x8 kf[KS], vf[KS];
  if (run_a) {
    load_frags<DH>(kb, a.ldk, k0, a.S, lane, kf);
    load_frags<DH>(vb, a.ldv, k0, a.S, lane, vf);
  }
  float lse_v = 0.f, d_v = 0.f;
  if (tid < RMAX && tid < a.T) {
    lse_v = a.lse[(long)bh * a.T + tid];
    if (!own_d) d_v = a.dvec[(long)bh * a.T + tid];
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int f = tid + NT * i, row = f / CH, ch = f - row * CH;
    if (row < Tt) {  // (images hold Tt / St rows; an iteration may reach past them)
      *reinterpret_cast<uint4*>(qimg + I::off(row, ch)) = rq[i];
      *reinterpret_cast<uint4*>(doimg + I::off(row, ch)) = rd[i];
      if (own_d) {
        float s = dot_bf16x8(rd[i], ro[i]);
        if (CH == 16) s = row16_sum(s);
        else {
#pragma unroll
          for (int m = CH / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        }
        if (ch == 0) dsh[row] = s;
      }
    }
    if (row < St) *reinterpret_cast<uint4*>(kimg + I::off(row, ch)) = rk[i];
  }
  if (tid < RMAX) {
    lsh[tid] = lse_v;
    if (!own_d) dsh[tid] = d_v;
  }
  ATTN_STAMP(1);  // loads landed, images written
  __syncthreads();
  ATTN_STAMP(2);

  // ---- phase A: keys on the columns; dK, dV; dS^T to LDS ----------------------------------------------------------
  const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  f32x4 dk[DT], dv[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) dk[d] = dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (run_a) {
    const int qbeg = a.causal ? (k0 & ~31) : 0;  // queries in front of the block's first key see none of it
    for (int qt = qbeg; qt < a.T; qt += 32) {
      f32x4 x[2], dp[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        x[t2] = dp[t2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          x[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::row_frag(qimg, qt + 16 * t2, ks, lane), kf[ks], x[t2], 0, 0, 0);
          dp[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::row_frag(doimg, qt + 16 * t2, ks, lane), vf[ks], dp[t2], 0, 0, 0);
        }
      }
      float pd[8], ds[8], keep[8];
      const uint64_t d0 = ((uint64_t)bh * a.T + qt + 4 * g) * (uint64_t)a.ld_drop + ki;  // dropout index of element 0
      if (a.drop_p > 0.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          keep[e] = drop_scale(a.seed, d0 + (uint64_t)((16 * (e >> 2) + (e & 3)) * a.ld_drop), a.drop_p, inv_keep);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) keep[e] = 1.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int qq = qt + 16 * (e >> 2) + 4 * g + (e & 3);
        const bool ok = key_ok && qq < a.T && (!a.causal || ki <= qq);
        const float p = ok ? __expf(x[e >> 2][e & 3] * a.scale - lsh[qq]) : 0.f;
        pd[e] = p * keep[e];
        ds[e] = p * (keep[e] * dp[e >> 2][e & 3] - dsh[qq]);
      }
      const bf16x8 pf = pack_frag(pd), sf = pack_frag(ds);
      {  // dS^T[key][queries qt + 4g .. + 3] and [.. + 16 ..]: two 8-byte stores
        union { bf16x8 f; uint4 u; } cv;
        cv.f = sf;
        const int c0 = qt + 4 * g;
        *reinterpret_cast<uint2*>(dst + IS::off(ki, c0 >> 3) + ((g & 1) << 3)) = make_uint2(cv.u.x, cv.u.y);
        *reinterpret_cast<uint2*>(dst + IS::off(ki, (c0 + 16) >> 3) + ((g & 1) << 3)) = make_uint2(cv.u.z, cv.u.w);
      }
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::tr_frag(doimg, qt, d, lane), pf, dv[d], 0, 0, 0);
        dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::tr_frag(qimg, qt, d, lane), sf, dk[d], 0, 0, 0);
      }
    }
  }
  ATTN_STAMP(3);  // phase A loop done (wave 0)
  __syncthreads();
  ATTN_STAMP(5);

  // ---- phase B: queries on the columns; dQ ------------------------------------------------------------------------
  if (wave < 4 * ((a.T + 63) / 64)) {
    const int q0 = wave * 16, qi = q0 + (lane & 15);
    f32x4 dq[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kmax = q0 < a.T ? (a.causal ? min(klim, q0 + 16) : klim) : 0;
    for (int kt = 0; kt < kmax; kt += 32) {
      const bf16x8 sf = IS::tr_frag(dst, kt, wave, lane);
#pragma unroll
      for (int d = 0; d < DT; ++d)
        dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::tr_frag(kimg, kt, d, lane), sf, dq[d], 0, 0, 0);
    }
    ATTN_STAMP(6);  // dQ products done
    const bool q_ok = qi < a.T;
    const long qo = ((long)b * a.T + min(qi, a.T - 1)) * a.ldq + h * DH;
    const long slot = a.db_part ? ((long)b * ((a.T + 63) / 64) * 4 + wave) * ((long)a.H * DH) : 0;
    store_grad<DT>(dq, a.scale, q_ok, a.dq ? a.dq + qo : nullptr, a.dqh ? a.dqh + qo : nullptr,
                   a.dbq ? a.dbq + slot + h * DH : nullptr, lane, a.db_part != 0);
    ATTN_STAMP(4);  // dQ stored
  }
  // dK, dV leave AFTER phase B: the barrier in front of it is then reached as soon as a wave's dS^T rows are in LDS, and the
  // gradient stores of the fast waves no longer sit in front of everybody's phase B (stamps: 3.9 k cycles of stores + 3 k of
  // waiting at the barrier per workgroup in the first form)
  // (bias-gradient partial slots: the two-pass kernels' numbering -- blocks of 64 columns, 4 waves each)
  if (wave < 4 * ((a.S + 63) / 64)) {
    const bool on = ki < a.S;
    const long ko = ((long)b * a.S + min(ki, a.S - 1)) * a.ldk + h * DH;
    const long vo = ((long)b * a.S + min(ki, a.S - 1)) * a.ldv + h * DH;
    const long slot = a.db_part ? ((long)b * ((a.S + 63) / 64) * 4 + wave) * ((long)a.H * DH) : 0;
    store_grad<DT>(dk, a.scale, on, a.dk ? a.dk + ko : nullptr, a.dkh ? a.dkh + ko : nullptr,
                   a.dbk ? a.dbk + slot + h * DH : nullptr, lane, a.db_part != 0);
    store_grad<DT>(dv, 1.f, on, a.dv ? a.dv + vo : nullptr, a.dvh ? a.dvh + vo : nullptr,
                   a.dbv ? a.dbv + slot + h * DH : nullptr, lane, a.db_part != 0);
  }
  ATTN_STAMP(7);
}

// Short-sequence forward (S <= 128, T <= 128): one workgroup of 8 waves per (batch, head) pair; K and V go to LDS ONCE
// (dual images, 64 KB at DH 128), every wave then walks the key tiles for its 16 queries without a barrier -- the
// streaming kernel above re-stages every 32-key tile per 64-query block behind two barriers.  Same arithmetic per tile.
inline int short_fwd_lds_bytes(int dh, int S) { return 2 * ((S + 31) & ~31) * dh * 2; }

template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void flash_fwd_short_kernel(AttnArgs a) {
  constexpr int KS = DH / 32, DT = DH / 16, NT = 64 * NW, CH = DH / 8;
  constexpr int NI = 16 * NW * CH / NT, RPI = NT / CH;
  if (a.T <= 0 || a.S <= 0) return;
  using I = Img2<DH>;
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  unsigned char* kimg = smem;
  unsigned char* vimg = smem + ((a.S + 31) & ~31) * I::PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int bh = blockIdx.x;
  const int b = bh / a.H, h = bh - b * a.H;
  const bf16_t* qb = a.q + (long)b * a.T * a.ldq + h * DH;
  const bf16_t* kb = a.k + (long)b * a.S * a.ldk + h * DH;
  const bf16_t* vb = a.v + (long)b * a.S * a.ldv + h * DH;
  const int klim = a.klen ? min((int)a.klen[b], a.S) : a.S;
  const int St = (klim + 31) & ~31;  // (keys past the longest valid one are never read)
  uint4 rk[NI], rv[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int f = tid + NT * i, row = f / CH, ch = f - row * CH;
    rk[i] = rv[i] = make_uint4(0, 0, 0, 0);
    if (RPI * i < St) {
      const long r = min(row, a.S - 1);
      const uint4 vk = *reinterpret_cast<const uint4*>(kb + r * a.ldk + ch * 8);
      const uint4 vv = *reinterpret_cast<const uint4*>(vb + r * a.ldv + ch * 8);
      if (row < a.S) { rk[i] = vk; rv[i] = vv; }
    }
  }
  const int q0 = wave * 16, qi = q0 + (lane & 15);
  const bool run = q0 < a.T;
  bf16x8 qf[KS];
  if (run) load_frags<DH>(qb, a.ldq, q0, a.T, lane, qf);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int f = tid + NT * i, row = f / CH, ch = f - row * CH;
    if (row < St) {
      *reinterpret_cast<uint4*>(kimg + I::off(row, ch)) = rk[i];
      *reinterpret_cast<uint4*>(vimg + I::off(row, ch)) = rv[i];
    }
  }
  __syncthreads();
  if (!run) return;
  const int kmax = a.causal ? min(klim, q0 + 16) : klim;
  f32x4 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  const uint64_t drow = ((uint64_t)bh * a.T + qi) * (uint64_t)a.ld_drop;
  for (int kt = 0; kt < kmax; kt += 32) {
    f32x4 x[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      x[t2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        x[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::row_frag(kimg, kt + 16 * t2, ks, lane), qf[ks], x[t2], 0, 0, 0);
    }
    float sc[8];
    float mt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = kt + 16 * (e >> 2) + 4 * g + (e & 3);
      const bool ok = key < klim && (!a.causal || key <= qi);
      sc[e] = ok ? x[e >> 2][e & 3] * a.scale : -INFINITY;
      mt = fmaxf(mt, sc[e]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 16));
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float mn = fmaxf(m, mt);
    const float mu = mn == -INFINITY ? 0.f : mn;
    const float alpha = __expf(m - mu);  // m = -inf -> 0
    float p[8], ps = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p[e] = __expf(sc[e] - mu);
      ps += p[e];
    }
    if (a.drop_p > 0.f) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        p[e] *= drop_scale(a.seed, drow + (uint64_t)(kt + 16 * (e >> 2) + 4 * g + (e & 3)), a.drop_p, inv_keep);
    }
    l = l * alpha + ps;
    m = mn;
    const bf16x8 pf = pack_frag(p);
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      o[d] *= alpha;
      o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::tr_frag(vimg, kt, d, lane), pf, o[d], 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float inv_l = l > 0.f ? 1.f / l : 0.f;
  if (qi < a.T) {
    const long ro = ((long)b * a.T + qi) * ((long)a.H * DH) + h * DH;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const float4 v = make_float4(o[d][0] * inv_l, o[d][1] * inv_l, o[d][2] * inv_l, o[d][3] * inv_l);
      *reinterpret_cast<float4*>(a.o + ro + 16 * d + 4 * g) = v;
      if (a.oh) *reinterpret_cast<uint2*>(a.oh + ro + 16 * d + 4 * g) = pack_bf16x4(v.x, v.y, v.z, v.w);
    }
    if (g == 0 && a.lse) a.lse[(long)bh * a.T + qi] = l > 0.f ? m + __logf(l) : -INFINITY;
  }
}

// waves per workgroup (16 columns each): 2 -> T/32 x B*H workgroups, several resident per CU, so the
// barrier / global-load latency of one overlaps the MFMAs of another (sequences here are 100-750 long)
constexpr int ANW = 4;  // (measured in round 2: 4 > 2 > 1 on the bench step; the switch and the other instantiations are gone)
constexpr int attn_nw() { return ANW; }

// the short-sequence kernels use more than 64 KB of dynamic LDS: set the attribute once per instantiation
bool short_configure() {
  static int ok = -1;
  if (ok < 0) {
    auto set = [](auto kern, int bytes) {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    };
    ok = set(flash_fwd_short_kernel<128, 8>, short_fwd_lds_bytes(128, 128)) && set(flash_fwd_short_kernel<64, 8>, short_fwd_lds_bytes(64, 128)) &&
         set(flash_fwd_short_kernel<128, 4>, short_fwd_lds_bytes(128, 64)) && set(flash_fwd_short_kernel<64, 4>, short_fwd_lds_bytes(64, 64)) &&
         set(flash_bwd_short_kernel<128, 8>, short_lds_bytes(128, 128, 128)) && set(flash_bwd_short_kernel<64, 8>, short_lds_bytes(64, 128, 128)) &&
         set(flash_bwd_short_kernel<128, 4>, short_lds_bytes(128, 64, 64)) && set(flash_bwd_short_kernel<64, 4>, short_lds_bytes(64, 64, 64)) ? 1 : 0;
  }
  return ok == 1;
}
// S2ST_ATTN_SHORT=0 (read per call: the A/B test flips it): the streaming / two-pass kernels at every length
bool short_enabled() {
  const char* e = s2st_env_str("S2ST_ATTN_SHORT");
  return !(e && atoi(e) == 0);
}
bool short_fwd_enabled() { return short_enabled(); }

bool attn_args_ok(const s2st_attn_args& p) {
  auto al = [](const void* x, int n) { return ((uintptr_t)x % n) == 0; };
  return (p.dh == 64 || p.dh == 128) && p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && al(p.q, 16) &&
         al(p.k, 16) && al(p.v, 16) && p.B > 0 && p.H > 0 && p.T > 0 && p.S > 0;
}

AttnArgs to_args(const s2st_attn_args& p) {
  AttnArgs a{};
  a.q = p.q; a.k = p.k; a.v = p.v;
  a.ldq = p.ldq; a.ldk = p.ldk; a.ldv = p.ldv;
  a.o = p.o; a.oh = p.oh; a.lse = p.lse; a.klen = p.klen;
  a.B = p.B; a.H = p.H; a.T = p.T; a.S = p.S; a.causal = p.causal;
  a.scale = p.scale; a.drop_p = p.drop_p; a.seed = p.seed; a.ld_drop = p.ld_drop;
  a.doh = p.doh; a.dq = p.dq; a.dk = p.dk; a.dv = p.dv;
  a.dqh = p.dqh; a.dkh = p.dkh; a.dvh = p.dvh; a.dbq = p.dbq; a.dbk = p.dbk; a.dbv = p.dbv;
  a.plain_order = 0;  // (XCD-aware block order: round 3's A/B -- 68 -> 18 MB fetched per backward launch -- is closed)
  return a;
}

}  // namespace

int s2st_flash_attn_supported(int dh) { return dh == 64 || dh == 128; }

// empty launches of every instantiation: code objects resident before the first timed step
int s2st_flash_attn_preload(hipStream_t st) {
  AttnArgs a{};
  auto go = [&](auto nwc) {
    constexpr int NW = decltype(nwc)::value;
    S2ST_LAUNCH((flash_fwd_kernel<128, NW>), dim3(1, 1), dim3(64 * NW), 0, st, a);
    S2ST_LAUNCH((flash_fwd_kernel<64, NW>), dim3(1, 1), dim3(64 * NW), 0, st, a);
    S2ST_LAUNCH((flash_bwd_kv_kernel<128, NW>), dim3(1, 1), dim3(64 * NW), 0, st, a);
    S2ST_LAUNCH((flash_bwd_kv_kernel<64, NW>), dim3(1, 1), dim3(64 * NW), 0, st, a);
    S2ST_LAUNCH((flash_bwd_q_kernel<128, NW>), dim3(1, 1), dim3(64 * NW), 0, st, a);
    S2ST_LAUNCH((flash_bwd_q_kernel<64, NW>), dim3(1, 1), dim3(64 * NW), 0, st, a);
    S2ST_LAUNCH((flash_bwd_kernel<128, NW>), dim3(1, 1), dim3(64 * NW), 0, st, a, 1);
    S2ST_LAUNCH((flash_bwd_kernel<64, NW>), dim3(1, 1), dim3(64 * NW), 0, st, a, 1);
  };
  go(std::integral_constant<int, ANW>{});
  if (!short_configure()) return S2ST_ERR_LAUNCH;
  S2ST_LAUNCH((flash_fwd_short_kernel<128, 8>), dim3(1), dim3(512), 0, st, a);
  S2ST_LAUNCH((flash_fwd_short_kernel<64, 8>), dim3(1), dim3(512), 0, st, a);
  S2ST_LAUNCH((flash_fwd_short_kernel<128, 4>), dim3(1), dim3(256), 0, st, a);
  S2ST_LAUNCH((flash_fwd_short_kernel<64, 4>), dim3(1), dim3(256), 0, st, a);
  S2ST_LAUNCH((flash_bwd_short_kernel<128, 8>), dim3(1), dim3(512), 0, st, a);
  S2ST_LAUNCH((flash_bwd_short_kernel<64, 8>), dim3(1), dim3(512), 0, st, a);
  S2ST_LAUNCH((flash_bwd_short_kernel<128, 4>), dim3(1), dim3(256), 0, st, a);
  S2ST_LAUNCH((flash_bwd_short_kernel<64, 4>), dim3(1), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_flash_attn_fwd(const s2st_attn_args* p, hipStream_t st) {
  if (!p || !attn_args_ok(*p) || !p->o || !p->lse) return S2ST_ERR_ARG;
  AttnArgs a = to_args(*p);
  if (p->T <= SHORT_MAX && p->S <= SHORT_MAX && short_fwd_enabled() && short_configure()) {
    const double fl = 4.0 * p->B * p->H * (double)p->T * p->S * p->dh * (p->causal ? 0.5 : 1.0);
    const dim3 grid(p->B * p->H);
    const int lds = short_fwd_lds_bytes(p->dh, p->S);
    const bool w4 = p->T <= 64 && p->S <= 64;
    if (p->dh == 128 && w4) s2st_launch("flash_fwd_short_kernel<128, 4>", fl, 0.0, flash_fwd_short_kernel<128, 4>, grid, dim3(256), lds, st, a);
    else if (p->dh == 128) s2st_launch("flash_fwd_short_kernel<128, 8>", fl, 0.0, flash_fwd_short_kernel<128, 8>, grid, dim3(512), lds, st, a);
    else if (w4) s2st_launch("flash_fwd_short_kernel<64, 4>", fl, 0.0, flash_fwd_short_kernel<64, 4>, grid, dim3(256), lds, st, a);
    else s2st_launch("flash_fwd_short_kernel<64, 8>", fl, 0.0, flash_fwd_short_kernel<64, 8>, grid, dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
  }
  auto go = [&](auto nwc) {
    constexpr int NW = decltype(nwc)::value;
    dim3 grid((p->T + 16 * NW - 1) / (16 * NW), p->B * p->H);
    // as-launched FLOPs over the padded T x S rectangle (QK^T and PV), halved for the causal form
    const double fl = 4.0 * p->B * p->H * (double)p->T * p->S * p->dh * (p->causal ? 0.5 : 1.0);
    if (p->dh == 128) s2st_launch("flash_fwd_kernel<128>", fl, 0.0, flash_fwd_kernel<128, NW>, grid, dim3(64 * NW), 0, st, a);
    else s2st_launch("flash_fwd_kernel<64>", fl, 0.0, flash_fwd_kernel<64, NW>, grid, dim3(64 * NW), 0, st, a);
  };
  go(std::integral_constant<int, ANW>{});
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

// dO (fp32, for D) + its bf16 copy; o / lse from the forward; writes dq, dk, dv (fp32, overwrite)
// phase 0: everything; 1: only the D vector; 2: only dK,dV; 3: only dQ (the caller may overlap 2 and 3)
// floats of scratch for the bias gradients in partial-sum form (0 when no bias gradient is asked for)
long s2st_flash_attn_db_scratch_floats(const s2st_attn_args* p) {
  if (!p || (!p->dbq && !p->dbk && !p->dbv)) return 0;
  const int nw = attn_nw();
  const long C = (long)p->H * p->dh;
  const long sq = (long)p->B * ((p->T + 16 * nw - 1) / (16 * nw)) * nw, sk = (long)p->B * ((p->S + 16 * nw - 1) / (16 * nw)) * nw;
  return (sq + 2 * sk) * C;
}

void s2st_flash_attn_db_layout(const s2st_attn_args* p, int* slots_q, int* slots_k) {
  const int nw = attn_nw();
  *slots_q = p->B * ((p->T + 16 * nw - 1) / (16 * nw)) * nw;
  *slots_k = p->B * ((p->S + 16 * nw - 1) / (16 * nw)) * nw;
}

int s2st_flash_attn_bwd(const s2st_attn_args* p, const float* dO, float* dvec_scratch, hipStream_t st, int phase,
                        float* db_part) {
  if (!p || !attn_args_ok(*p) || !p->doh || (!p->dq && !p->dqh) || (!p->dk && !p->dkh) || (!p->dv && !p->dvh) || !p->lse)
    return S2ST_ERR_ARG;
  // D = rowsum(dO * O): inside the backward kernels from the bf16 copies when the forward left one of O (no separate
  // launch; without a bf16 copy of O: the fp32 row kernel first)
  const bool own_d = p->oh && phase == 0;
  if (!own_d && (!dO || !dvec_scratch || !p->o)) return S2ST_ERR_ARG;
  AttnArgs a = to_args(*p);
  a.dvec = own_d ? nullptr : dvec_scratch;
#ifdef S2ST_ATTN_STAMP
  a.stamp = reinterpret_cast<long*>(dvec_scratch);
#endif
  // bias gradients without atomics: the kernels write per-(block, wave) partial sums, folded in slot order below
  const int nwp = attn_nw();
  const long Cp = (long)p->H * p->dh;
  const long slots_q = (long)p->B * ((p->T + 16 * nwp - 1) / (16 * nwp)) * nwp, slots_k = (long)p->B * ((p->S + 16 * nwp - 1) / (16 * nwp)) * nwp;
  const bool part = db_part && phase == 0 && (p->dbq || p->dbk || p->dbv);
  if (part) {
    a.db_part = 1;
    a.dbq = p->dbq ? db_part : nullptr;
    a.dbk = p->dbk ? db_part + slots_q * Cp : nullptr;
    a.dbv = p->dbv ? db_part + (slots_q + slots_k) * Cp : nullptr;
  }
  // T, S <= 128: one workgroup per (batch, head) pair, every operand loaded once, the scores computed once
  if (phase == 0 && p->T <= SHORT_MAX && p->S <= SHORT_MAX && short_enabled() && short_configure()) {
    if (!own_d) {  // (D from the fp32 row kernel: the no-bf16-O form)
      const long rows_ = (long)p->B * p->T * p->H;
      if (p->dh == 128)
        S2ST_LAUNCH(attn_dvec_kernel<128>, dim3((unsigned)((rows_ * 32 + 255) / 256)), dim3(256), 0, st, dO, (const float*)p->o,
                    dvec_scratch, p->B, p->H, p->T);
      else
        S2ST_LAUNCH(attn_dvec_kernel<64>, dim3((unsigned)((rows_ * 16 + 255) / 256)), dim3(256), 0, st, dO, (const float*)p->o,
                    dvec_scratch, p->B, p->H, p->T);
    }
    // as-launched FLOPs: S, dP, dV, dK, dQ over the padded T x S rectangle (the causal form skips whole tiles only)
    const double fl = 5.0 * 2.0 * p->B * p->H * (double)p->T * p->S * p->dh * (p->causal ? 0.5 : 1.0);
    const dim3 grid(p->B * p->H);
    const int lds = short_lds_bytes(p->dh, p->T, p->S);
    const bool w4 = p->T <= 64 && p->S <= 64;
    if (p->dh == 128 && w4) s2st_launch("flash_bwd_short_kernel<128, 4>", fl, 0.0, flash_bwd_short_kernel<128, 4>, grid, dim3(256), lds, st, a);
    else if (p->dh == 128) s2st_launch("flash_bwd_short_kernel<128, 8>", fl, 0.0, flash_bwd_short_kernel<128, 8>, grid, dim3(512), lds, st, a);
    else if (w4) s2st_launch("flash_bwd_short_kernel<64, 4>", fl, 0.0, flash_bwd_short_kernel<64, 4>, grid, dim3(256), lds, st, a);
    else s2st_launch("flash_bwd_short_kernel<64, 8>", fl, 0.0, flash_bwd_short_kernel<64, 8>, grid, dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
  }
  const long rows = (long)p->B * p->T * p->H;
  if (phase > 1 || own_d) {
  } else if (p->dh == 128)
    S2ST_LAUNCH(attn_dvec_kernel<128>, dim3((unsigned)((rows * 32 + 255) / 256)), dim3(256), 0, st, dO,
                       (const float*)p->o, dvec_scratch, p->B, p->H, p->T);
  else
    S2ST_LAUNCH(attn_dvec_kernel<64>, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, st, dO,
                       (const float*)p->o, dvec_scratch, p->B, p->H, p->T);
  auto go = [&](auto nwc) {
    constexpr int NW = decltype(nwc)::value;
    dim3 gk((p->S + 16 * NW - 1) / (16 * NW), p->B * p->H), gq((p->T + 16 * NW - 1) / (16 * NW), p->B * p->H);
    const bool kv = phase == 0 || phase == 2, qq = phase == 0 || phase == 3;
    // as-launched FLOPs: dK,dV pass = S^T recompute + dP + dV + dK (4 products), dQ pass = S + dP + dQ (3 products)
    const double f1 = 2.0 * p->B * p->H * (double)p->T * p->S * p->dh * (p->causal ? 0.5 : 1.0);
    if (kv && qq) {
      dim3 g2(gk.x + gq.x, gk.y);
      if (p->dh == 128) s2st_launch("flash_bwd_kernel<128>", 7 * f1, 0.0, flash_bwd_kernel<128, NW>, g2, dim3(64 * NW), 0, st, a, (int)gk.x);
      else s2st_launch("flash_bwd_kernel<64>", 7 * f1, 0.0, flash_bwd_kernel<64, NW>, g2, dim3(64 * NW), 0, st, a, (int)gk.x);
    } else if (p->dh == 128) {
      if (kv) s2st_launch("flash_bwd_kv_kernel<128>", 4 * f1, 0.0, flash_bwd_kv_kernel<128, NW>, gk, dim3(64 * NW), 0, st, a);
      if (qq) s2st_launch("flash_bwd_q_kernel<128>", 3 * f1, 0.0, flash_bwd_q_kernel<128, NW>, gq, dim3(64 * NW), 0, st, a);
    } else {
      if (kv) s2st_launch("flash_bwd_kv_kernel<64>", 4 * f1, 0.0, flash_bwd_kv_kernel<64, NW>, gk, dim3(64 * NW), 0, st, a);
      if (qq) s2st_launch("flash_bwd_q_kernel<64>", 3 * f1, 0.0, flash_bwd_q_kernel<64, NW>, gq, dim3(64 * NW), 0, st, a);
    }
  };
  go(std::integral_constant<int, ANW>{});
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

// db[c] += sum over slots of the partial sums s2st_flash_attn_bwd(..., db_part) left, in slot order: one launch for the
// three projections (blockIdx.y = q / k / v); any stream ordered behind the backward kernels
__global__ __launch_bounds__(256) void attn_db_fold_kernel(const float* __restrict__ part, int slots_q, int slots_k, int C,
                                                           float* __restrict__ dbq, float* __restrict__ dbk,
                                                           float* __restrict__ dbv) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int which = blockIdx.y;
  float* out = which == 0 ? dbq : (which == 1 ? dbk : dbv);
  if (c >= C || !out) return;
  const float* src = part + (which == 0 ? 0 : (which == 1 ? (long)slots_q * C : (long)(slots_q + slots_k) * C));
  const int slots = which == 0 ? slots_q : slots_k;
  float s = 0.f;
  for (int sb = 0; sb < slots; sb += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(long)min(sb + j, slots - 1) * C + c];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (sb + j < slots) s += v[j];
  }
  out[c] += s;
}

int s2st_flash_attn_db_fold(const s2st_attn_args* p, const float* db_part, hipStream_t st) {
  if (!p || !db_part || (!p->dbq && !p->dbk && !p->dbv)) return 0;
  const int nw = attn_nw();
  const int C = p->H * p->dh;
  const int slots_q = p->B * ((p->T + 16 * nw - 1) / (16 * nw)) * nw, slots_k = p->B * ((p->S + 16 * nw - 1) / (16 * nw)) * nw;
  S2ST_LAUNCH(attn_db_fold_kernel, dim3((C + 255) / 256, 3), dim3(256), 0, st, db_part, slots_q, slots_k, C, p->dbq, p->dbk,
              p->dbv);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
