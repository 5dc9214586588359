// bf16 GEMM, 256 x 256 tile over a ring of four 32-deep half-tiles (round 6; round 5's form kept two 64-deep buffers):
// C(m, n) = epi(alpha * sum_k A(m, k) B(n, k)), both operands K-contiguous (the forward linear layers; data gradients
// against the pre-transposed weight copies; HuBERT).
//
// Why a third form.  The 128-row ring kernels need 64 B/clk/CU of operand fill to keep the MFMAs busy (32 KB per 64-deep
// K-step for 2.1 MFLOP = 512 MFMA clocks) and a CU takes in ~34 B/clk (MI355X_MICROARCH.md, gather / ldsdma-fill rows;
// measured here: 950 clocks per K-step): they are fill-bound at about half the MFMA rate, whatever their pipelining.  A
// 256 x 256 tile moves 64 KB per K-step for 8.4 MFLOP = 2048 MFMA clocks, i.e. needs 32 B/clk/CU -- the first tile shape
// that CAN be matrix-bound on this chip (cdna_hip_programming.md section 5, "The 256^2 8-phase template").  It is picked
// where a product has enough 256 x 256 tiles to occupy the chip: HuBERT's projections (M = 9.6 k ... 307 k rows).
//
// Structure (8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 MFMA tiles of 16 x 16, 128 accumulator registers):
//   * LDS = a ring of FOUR half-tile slots of 32 KB (A 256 rows x 32 k, B 256 rows x 32 k; 64-byte rows), filled by LDS-DMA:
//     4 wave-instructions of 1 KB (16 rows x 64 B) per wave and half-tile.  Round 5's two 64 KB buffers had ONE K-tile in
//     flight for ~3/4 of a K-tile's time (~14 B/clk of fill per CU by Little's law, 2.0 us per K-tile against the matrix
//     pipe's 0.85); here THREE half-tiles (96 KB) are in flight all the time -- the 128-row ring's depth -- and a DMA is
//     issued three half-tile periods before its data is needed;
//   * the 64-byte-row image: 16-byte chunk c of row r sits in slot c ^ g((r >> 2) & 3), g = (0, 3, 2, 1): conflict-free for
//     the four 16-lane groups ds_read_b128 is serviced in (MI355X_MICROARCH.md, LDS table: a group reads rows {0-3, 12-15}
//     at chunk c and rows {4-11} at chunk c + 1, or the reverse); the DMA applies the permutation on the SOURCE address
//     (lane l of a piece -> row l >> 2, slot l & 3, source chunk (l & 3) ^ g(l >> 4));
//   * a half-tile is multiplied as four QUADRANTS of the wave tile (64 x 32 each, 8 MFMAs); even half-tiles in the order
//     (m-half, n-half) = (0,0) (0,1) (1,1) (1,0), odd ones (0,1) (0,0) (1,0) (1,1): consecutive quadrants share one
//     operand's fragments, and the next half-tile starts with the B half the current one ends with its OTHER B half on --
//     so every fragment set (A m-half 0 / 1: 16 registers each, B n-half 0 / 1: 8 each) is re-filled right behind the MFMAs
//     that used it last, with a whole quadrant (8 MFMAs, 128 matrix-pipe clocks) or more in front of its next use: no read
//     is exposed, not even behind the barrier, and no fragment is read twice;
//   * ONE barrier per half-tile, in its MIDDLE (after the second quadrant).  In front of it a wave waits for its own
//     fragment reads of this half-tile (all issued during the first quadrant: lgkmcnt(0), a formality) and for its own DMA
//     pieces of the NEXT half-tile (counted vmcnt: two younger half-tiles stay in flight).  Behind it (i) every wave's
//     pieces of the next half-tile have landed -- its fragments are read during quadrants three and four, i.e. a read of a
//     slot follows the barrier that follows the wait (cdna_hip_programming.md: "read a staged buffer one phase AFTER the
//     wait that retires it") -- and (ii) nobody reads this half-tile's slot any more: the DMA of half-tile h + 4 goes there.
// K tail: K is processed in whole 64-deep steps; source offsets are clamped into the padded row and the k >= K part of
// the last one or two half-tiles is zeroed in LDS behind the barrier that publishes them (uniform branch, tail only).
// Summation order over k per accumulator: ascending 32-deep steps -- the 128-row ring kernels' order (bit-equal results).
// Epilogues are those of the ring kernels.
// Replaces F.linear of fairseq/modules/transformer_layer.py:140-162, multihead_attention.py:170-192 and
// fairseq/models/wav2vec/wav2vec2.py:736-814, 915-1016 (HuBERT projections / convolutions as GEMMs) in fast mode.
#include "gemm_bf16_tile.h"

namespace {

constexpr int P4_BM = 256, P4_BN = 256, P4_NW = 8, P4_WGN = 4;
constexpr int P4_HK = 32;                      // k depth of a half-tile
constexpr int P4_HALF = 256 * 2 * P4_HK;       // one operand's half-tile image: 256 rows x 64 B = 16 KB
constexpr int P4_SLOT = 2 * P4_HALF;           // A + B
constexpr int P4_NS = 4;                       // ring slots
constexpr int P4_LDS = P4_NS * P4_SLOT;        // 128 KB

__device__ __forceinline__ int p4_g(int q) { return q ^ ((q & 1) << 1); }  // (0, 1, 2, 3) -> (0, 3, 2, 1)

// zero k >= kv of a landed half-tile image (256 rows x 64 B, the swizzle above); kv <= 0 clears it
__device__ __forceinline__ void p4_sanitize(unsigned char* img, int kv, int tid) {
  for (int f = tid; f < 256 * 4; f += 64 * P4_NW) {
    const int row = f >> 2, slot = f & 3;
    const int nv = kv - 8 * (slot ^ p4_g((row >> 2) & 3));
    if (nv < 8) {
      uint4* q = reinterpret_cast<uint4*>(img + row * 64 + slot * 16);
      *q = mask_tail(*q, nv > 0 ? nv : 0);
    }
  }
}

__device__ __forceinline__ void gemm_p4_tile(const GemmArgs& g, int id, const int nwg, const int by) {
  constexpr int BM = P4_BM, BN = P4_BN, WGN = P4_WGN;
  constexpr int WM = BM / 2, WN = BN / WGN, TM = WM / 16, TN = WN / 16;  // 128 x 64: 8 x 4 MFMA tiles
  static_assert(TM == 8 && TN == 4, "wave tile");
  HIP_DYNAMIC_SHARED(unsigned char, smem)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  if (nwg > 0) {  // XCD-aware tile order: ids that share an XCD (id % 8) own a contiguous run of tiles
    const int x = id & 7, q = nwg >> 3, r = nwg & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
  }
  const int tile_m = id / g.tiles_n, tile_n = id - tile_m * g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int zb = by / g.splitk, ks = by - zb * g.splitk;
  const int zq = zb / g.zdiv, zr = zb - zq * g.zdiv;
  const bf16_t* abase = reinterpret_cast<const bf16_t*>(g.A.p) + zq * g.A.zo + zr * g.A.zi;
  const bf16_t* bbase = reinterpret_cast<const bf16_t*>(g.B.p) + zq * g.B.zo + zr * g.B.zi;
  const int kbeg = ks * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nh = 2 * ((kend - kbeg + BK - 1) / BK);  // half-tiles (whole 64-deep steps: an even number)
  const bool tail = ((kend - kbeg) & (BK - 1)) != 0;  // (wave-uniform)

  // LDS-DMA: piece j (j < 2) of an operand's half-tile = rows [32 wave + 16 j, + 16) of the tile; lane l -> row l >> 2,
  // 16-byte slot l & 3, source chunk (l & 3) ^ g(l >> 4).  Plain row strides only (no conv window addressing: p4_pick
  // leaves those products to the ring kernels).  Row bases are kept as four pointers; the k offset is formed per issue.
  const int dchunk8 = 8 * ((lane & 3) ^ p4_g(lane >> 4));
  const int kpad = ((g.K + 7) & ~7) - 8;  // (loads stay inside the padded row: the K tail is zeroed in LDS)
  const bf16_t *pa[2], *pb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    pa[j] = abase + (long)min(m0 + wave * 32 + 16 * j + (lane >> 2), g.M - 1) * g.A.sp.ld;
    pb[j] = bbase + (long)min(n0 + wave * 32 + 16 * j + (lane >> 2), g.N - 1) * g.B.sp.ld;
  }
  auto issue = [&](int h) {  // the DMA of half-tile h into its slot
    unsigned char* s = smem + (h & (P4_NS - 1)) * P4_SLOT + wave * (32 * 64);
    const int ko = min(kbeg + h * P4_HK + dchunk8, kpad);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(pa[j] + ko), (lptr_t)(s + j * (16 * 64)), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(pb[j] + ko), (lptr_t)(s + P4_HALF + j * (16 * 64)), 16, 0, 0);
  };
  // own pieces of every half-tile but the `ahead` youngest have landed (4 DMA instructions per wave and half-tile)
  auto wait_dma = [&](int ahead) {
    if (ahead >= 3) S2ST_VMCNT(12);
    else if (ahead == 2) S2ST_VMCNT(8);
    else if (ahead == 1) S2ST_VMCNT(4);
    else S2ST_VMCNT(0);
  };
  // half-tile h has been published by a barrier: clear its k >= K part (tail products only), and publish that
  auto clean = [&](int h) {
    const int kv = kend - (kbeg + h * P4_HK);
    if (tail && kv < P4_HK) {
      unsigned char* s = smem + (h & (P4_NS - 1)) * P4_SLOT;
      p4_sanitize(s, kv, tid);
      p4_sanitize(s + P4_HALF, kv, tid);
      __syncthreads();
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment of 16 tile rows from row rt (rt % 16 == 0) of a half-tile image: lane l holds row l & 15, k = 8 (l >> 4) + j
  const int foff = (lane & 15) * 64 + (((lane >> 4) ^ p4_g((lane & 15) >> 2)) << 4);
  const int aoff = wm * WM * 64 + foff, boff = P4_HALF + wn * WN * 64 + foff;
  bf16x8 aP[4], aQ[4], bX[2], bY[2];  // A m-half 0 / 1, B n-half 0 / 1 of the wave tile
  // (reads by hand, s2st_asm.h: they stay in flight across MFMA groups, loop edges and the barrier; every use of a set is
  // ordered behind the counted wait that retires it by P4_USE_*; sched_barrier pins the instruction order)
#define P4_READ_A(D, S, R0)                                                                                              \
  D[0] = lds_read_b128_raw<((R0) + 0) * 64>((S) + aoff);                                                                 \
  D[1] = lds_read_b128_raw<((R0) + 16) * 64>((S) + aoff);                                                                \
  D[2] = lds_read_b128_raw<((R0) + 32) * 64>((S) + aoff);                                                                \
  D[3] = lds_read_b128_raw<((R0) + 48) * 64>((S) + aoff);
#define P4_READ_B(D, S, C0)                                                                                              \
  D[0] = lds_read_b128_raw<((C0) + 0) * 64>((S) + boff);                                                                 \
  D[1] = lds_read_b128_raw<((C0) + 16) * 64>((S) + boff);
#define P4_USE_A(D) lds_raw_fence(D[0]); lds_raw_fence(D[1]); lds_raw_fence(D[2]); lds_raw_fence(D[3]);
#define P4_USE_B(D) lds_raw_fence(D[0]); lds_raw_fence(D[1]);
#define P4_MFMA8(A_, B_, I0, J0)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                            \
      acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B_[j], A_[i], acc[(I0) + i][(J0) + j], 0, 0, 0);
#define P4_FENCE() __builtin_amdgcn_sched_barrier(0)
  // counted LDS waits (reads return in order): N = the reads issued AFTER the ones the next MFMA group consumes
#define P4_LGKM(N) __builtin_amdgcn_s_waitcnt(0xc07f | ((N) << 8))
  // the middle of half-tile H: its reads are done, the next one's pieces are in; publish both; refill this slot
#define P4_MIDDLE(H)                                                                                                     \
  P4_LGKM(0);                                                                                                            \
  if ((H) + 1 < nh) wait_dma(min((H) + 3, nh - 1) - ((H) + 1));                                                          \
  __builtin_amdgcn_s_barrier();                                                                                          \
  if ((H) + P4_NS < nh) issue((H) + P4_NS);                                                                              \
  if ((H) + 1 < nh) clean((H) + 1);                                                                                      \
  P4_FENCE();

  {  // prologue: four half-tiles on their way, the first one published, its first fragments requested
    const int pre = min(nh, P4_NS);
    for (int h = 0; h < pre; ++h) issue(h);
    wait_dma(pre - 1);
    __builtin_amdgcn_s_barrier();
    clean(0);
    P4_READ_A(aP, smem, 0)
    P4_READ_B(bX, smem, 0)
    P4_FENCE();
  }
  for (int h = 0; h < nh; h += 2) {
    const unsigned char* s0 = smem + (h & (P4_NS - 1)) * P4_SLOT;
    const unsigned char* s1 = smem + ((h + 1) & (P4_NS - 1)) * P4_SLOT;
    const unsigned char* s2 = smem + ((h + 2) & (P4_NS - 1)) * P4_SLOT;
    const bool more = h + 2 < nh;  // (wave-uniform, the same in every wave)
    // ---- even half-tile: (0,0) (0,1) | (1,1) (1,0) -----------------------------------------------------------------
    P4_READ_B(bY, s0, 32)
    P4_READ_A(aQ, s0, 64)
    P4_FENCE();
    P4_LGKM(6);
    P4_USE_A(aP) P4_USE_B(bX)
    P4_MFMA8(aP, bX, 0, 0)
    P4_FENCE();
    P4_LGKM(4);
    P4_USE_B(bY)
    P4_MFMA8(aP, bY, 0, 2)
    P4_FENCE();
    P4_MIDDLE(h)
    P4_USE_A(aQ)
    P4_READ_A(aP, s1, 0)
    P4_FENCE();
    P4_MFMA8(aQ, bY, 4, 2)
    P4_FENCE();
    P4_READ_B(bY, s1, 32)
    P4_FENCE();
    P4_MFMA8(aQ, bX, 4, 0)
    P4_FENCE();
    // ---- odd half-tile: (0,1) (0,0) | (1,0) (1,1) ------------------------------------------------------------------
    P4_READ_B(bX, s1, 0)
    P4_READ_A(aQ, s1, 64)
    P4_FENCE();
    P4_LGKM(6);
    P4_USE_A(aP) P4_USE_B(bY)
    P4_MFMA8(aP, bY, 0, 2)
    P4_FENCE();
    P4_LGKM(4);
    P4_USE_B(bX)
    P4_MFMA8(aP, bX, 0, 0)
    P4_FENCE();
    P4_MIDDLE(h + 1)
    P4_USE_A(aQ)
    if (more) { P4_READ_A(aP, s2, 0) }
    P4_FENCE();
    P4_MFMA8(aQ, bX, 4, 0)
    P4_FENCE();
    if (more) { P4_READ_B(bX, s2, 0) }
    P4_FENCE();
    P4_MFMA8(aQ, bY, 4, 2)
    P4_FENCE();
  }
#undef P4_USE_A
#undef P4_USE_B
#undef P4_MIDDLE
#undef P4_LGKM
#undef P4_MFMA8
#undef P4_READ_A
#undef P4_READ_B
#undef P4_FENCE
  // (the epilogue's addresses, masks and bias loads depend only on kernel arguments and the lane: left visible, the compiler
  // computes them ABOVE the K-loop and keeps them in registers the loop needs -- it then spills accumulators inside the
  // loop and waits for them with vmcnt(0), which also drains the DMA.  Opaque copies pin that work behind the loop.)
  int lane_e = lane, m0_e = m0, n0_e = n0;
  opaque_v(lane_e);
  opaque_s(m0_e);
  opaque_s(n0_e);
  // The wave tile's 128 rows go through the shared epilogue as two blocks of 64 (its 128-row instantiation: a wave there
  // owns rows m0 + 64 wm + 16 i, i < 4): the straight-line form loads a block's residual / old values first, and for all
  // eight row tiles at once that is another 128 registers next to the 128 accumulators -- the allocator then spills
  // accumulators across the K-loop.  (Masked data-gradient products, whose bias partial rows are indexed by the tile
  // height, stay on the 128-row forms: p4_pick.)
  {
    f32x4 blk[4][TN];  // (plain copies: register renaming, no code -- a cast of &acc[4] would put the array in memory)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) blk[i][j] = acc[i][j];
    gemm_epilogue<128, BN, WGN, true>(g, blk, m0_e + 64 * wm, n0_e, wm, wn, lane_e, zb, ks, zq, zr);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) blk[i][j] = acc[4 + i][j];
    gemm_epilogue<128, BN, WGN, true>(g, blk, m0_e + 64 * wm + 64, n0_e, wm, wn, lane_e, zb, ks, zq, zr);
  }
}

__global__ __launch_bounds__(64 * P4_NW) void gemm_bf16_p4_kernel(GemmArgs g) {
  gemm_p4_tile(g, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

double p4_flops(const GemmArgs& g) { return 2.0 * g.M * g.N * (double)g.K * g.batch; }
double p4_min_bytes(const GemmArgs& g) {
  const double mn = (double)g.M * g.N * g.batch;
  return 2.0 * g.batch * ((double)g.M * g.K + (double)g.N * g.K) + mn * ((g.C.p ? 4 : 0) + (g.C.h ? 2 : 0)) +
         mn * 4 * ((g.ep.accumulate ? 1 : 0) + (g.ep.resid ? 1 : 0));
}

}  // namespace

// g: prepared by s2st_gemm_bf16 (alignment flags, tiles_n for the 256 x 256 tile, kchunk / splitk, epilogue marks);
// both operands K-contiguous and 16-byte aligned (the caller checked)
int s2st_gemm_bf16_p4(const GemmArgs& g, dim3 grid, hipStream_t st) {
  if (!g.A.kmajor || !g.B.kmajor) return S2ST_ERR_ARG;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_p4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            P4_LDS) != hipSuccess)
      return -1;
    configured = true;
  }
  s2st_launch("gemm_bf16_p4_kernel<256, 256>", p4_flops(g), p4_min_bytes(g), gemm_bf16_p4_kernel, grid, dim3(64 * P4_NW), P4_LDS,
              st, g);
  return 0;
}

int s2st_gemm_bf16_p4_preload(hipStream_t st) {
  GemmArgs g{};
  g.A.dtype = g.B.dtype = S2ST_BF16;
  g.A.kmajor = g.B.kmajor = 1;
  g.splitk = 1; g.zdiv = 1; g.tiles_n = 1; g.batch = 1; g.kchunk = BK;
  const int rc = s2st_gemm_bf16_p4(g, dim3(1), st);
  return rc || hipGetLastError() != hipSuccess ? -1 : 0;
}
