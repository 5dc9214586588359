// bf16 GEMM, 256 x 256 tile, a K-tile's operands as four 16 KB REGIONS that are re-filled one per quadrant (round 6;
// round 5's form filled a whole 64 KB buffer in two bursts): C(m, n) = epi(alpha * sum_k A(m, k) B(n, k)), both operands
// K-contiguous (the forward linear layers; data gradients against the pre-transposed weight copies; HuBERT).
//
// Why a third form.  The 128-row ring kernels need 64 B/clk/CU of operand fill to keep the MFMAs busy (32 KB per 64-deep
// K-step for 2.1 MFLOP = 512 MFMA clocks) and a CU takes in ~34 B/clk (MI355X_MICROARCH.md, gather / ldsdma-fill rows;
// measured here: 950 clocks per K-step): they are fill-bound at about half the MFMA rate, whatever their pipelining.  A
// 256 x 256 tile moves 64 KB per K-step for 8.4 MFLOP = 2048 MFMA clocks, i.e. needs 32 B/clk/CU -- the first tile shape
// that CAN be matrix-bound on this chip (cdna_hip_programming.md section 5, "The 256^2 8-phase template").  It is picked
// where a product has enough 256 x 256 tiles to occupy the chip: HuBERT's projections (M = 9.6 k ... 307 k rows).
//
// What bound round 5's form (2.0 us per K-tile, 1.09 PFLOP/s on 4096^3): an LDS-DMA instruction occupies its wave until
// the CU's load path has taken it (~30 clocks per 1 KB piece at 34 B/clk), and that form had all eight waves issue four
// pieces in front of the MFMAs of the first quadrant and four in front of the second: twice per K-tile every wave of the CU
// sat ~960 clocks in DMA issue with the matrix pipe empty -- 1900 + 2048 clocks per K-tile.  (A deeper ring does not help
// by itself: four 32-deep half-tile slots with three in flight measured SLOWER, 0.93 PFLOP/s -- their pieces are 16 rows x
// 64 B, the half-line shape that doubles the load path's work; profiles/r06_p4_half_tile_ring_bench.txt.)  The fill has
// to be spread evenly over the K-tile, 2 pieces per wave and quadrant, behind MFMAs that are already issued.
//
// Structure (8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 MFMA tiles of 16 x 16, 128 accumulator registers):
//   * two K-tile images of 64 KB (A 256 x 64, B 256 x 64: the K-contiguous swizzled images of gemm_bf16_tile.h, 128-byte
//     rows = whole cache lines per DMA row), each made of four REGIONS of 128 rows: A_m0 / A_m1 = the rows of the first /
//     second 64 of every wave's 128 A rows, B_n0 / B_n1 = the first / second 32 of every wave's 64 B rows.  A region is 16
//     pieces of 8 rows x 128 B: two per wave;
//   * a K-tile is multiplied as four QUADRANTS of the wave tile (64 x 32 each, k-half 0 then 1: 16 MFMAs):
//     (m0, n0) -> (m0, n1) -> (m1, n1) -> (m1, n0); fragment registers: a[k-half][4] (the current A half, 32 registers),
//     bn0 / bn1[k-half][2] (both B halves, 32: n0 is used by the first and the last quadrant and is kept, not read
//     twice).  A set is re-filled right behind the MFMAs that used it last, with 8 MFMAs or more in front of its next use;
//   * every region is read from LDS within ONE quadrant (A_m0 and B_n0 of the next K-tile during this K-tile's last
//     quadrant, B_n1 during the first, A_m1 during the second), so one region per quadrant becomes free: the quadrant's
//     barrier is followed by the DMA of the same region TWO K-tiles ahead (2 pieces per wave, issued behind the quadrant's
//     first 8 MFMAs).  Issue order A_m0, B_n0, B_n1, A_m1 per K-tile = the order of first use; region n of that order is
//     issued behind barrier n - 7 and must have landed before barrier n - 2: five quadrants (1.25 K-tiles) of lead, and
//     four to five regions (64 - 80 KB) in flight all the time;
//   * ONE barrier per quadrant.  In front of it a wave waits for its own pieces of every region but the four youngest
//     (counted vmcnt(8)); behind it those regions are published -- their first read is at least one barrier later
//     (cdna_hip_programming.md: "read a staged buffer one phase AFTER the wait that retires it") -- and the region whose
//     fragments the PREVIOUS quadrant consumed (its reads were waited for there) may be overwritten;
//   * fragment reads by hand (s2st_asm.h) with counted lgkmcnt: up to twelve stay in flight across a barrier.
// K tail (K % 64 != 0): source offsets are clamped into the padded row; in front of the first read of the LAST K-tile
// everything is drained once and the k >= K part of its image is zeroed (uniform branch, tail products only).
// Summation order over k per accumulator: ascending 32-deep steps -- the 128-row ring kernels' order (bit-equal results).
// Epilogues are those of the ring kernels.
// Replaces F.linear of fairseq/modules/transformer_layer.py:140-162, multihead_attention.py:170-192 and
// fairseq/models/wav2vec/wav2vec2.py:736-814, 915-1016 (HuBERT projections / convolutions as GEMMs) in fast mode.
#include "gemm_bf16_tile.h"

namespace {

constexpr int P4_BM = 256, P4_BN = 256, P4_NW = 8, P4_WGN = 4;
constexpr int P4_OP = 256 * 128;        // one operand's K-tile image: 256 rows x 128 B = 32 KB
constexpr int P4_IMG = 2 * P4_OP;       // A + B
constexpr int P4_LDS = 2 * P4_IMG;      // 128 KB

template <int ABL>
__device__ __forceinline__ void gemm_p4_tile(const GemmArgs& g, int id, const int nwg, const int by) {
  constexpr int BM = P4_BM, BN = P4_BN, NW = P4_NW, WGN = P4_WGN;
  constexpr int WM = BM / 2, WN = BN / WGN, TM = WM / 16, TN = WN / 16;  // 128 x 64: 8 x 4 MFMA tiles
  static_assert(TM == 8 && TN == 4, "wave tile");
  typedef Dma<true, BM, NW> DA;  // (image geometry and the K-tail rule; the loads themselves are issued below)
  typedef Dma<true, BN, NW> DB;
  static_assert(DA::BYTES == P4_OP && DB::BYTES == P4_OP, "image size");
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
  const int nt = (kend - kbeg + BK - 1) / BK;
  const int nr = 4 * nt;                              // regions, in issue order: 4 u + (0 A_m0, 1 B_n0, 2 B_n1, 3 A_m1)
  const bool tail = ((kend - kbeg) & (BK - 1)) != 0;  // (wave-uniform)

  // LDS-DMA: a region = 128 tile rows = 16 pieces of 8 rows x 128 B, two per wave.  Piece j of wave w covers region rows
  // [16 w + 8 j, + 8); region row -> tile row: A_m0 rows 64 (w >> 2) * 2 + 16 (w & 3) ..., i.e. the first 64 of either wave
  // row's 128 (A_m1: + 64); B_n0 rows 64 (w >> 1) + 16 (w & 1) ..., the first 32 of a wave column's 64 (B_n1: + 32).  Lane l
  // -> row l >> 3 of the piece, 16-byte slot l & 7, source chunk (l & 7) ^ (l >> 3) (the image's swizzle, gemm_bf16_tile.h).
  // Addresses are formed on the fly from the lane's first row of either operand (plain row strides only: p4_pick leaves
  // conv-window products to the ring kernels).
  const int chunk8 = 8 * ((lane & 7) ^ (lane >> 3));
  const int kpad = ((g.K + 7) & ~7) - 8;  // (loads stay inside the padded row: the K tail is zeroed in LDS)
  const int ra0 = (wave >> 2) * 128 + (wave & 3) * 16, rb0 = (wave >> 1) * 64 + (wave & 1) * 16;  // tile rows of piece 0
  const long lda = g.A.sp.ld, ldb = g.B.sp.ld;
  auto issue = [&](int u, int r) {  // region r of K-tile u (r is a literal at every call site)
    const int ko = min(kbeg + u * BK + chunk8, kpad);
    unsigned char* img = smem + (u & 1) * P4_IMG;
    if (r == 0 || r == 3) {
      int row = ra0 + (r == 3 ? 64 : 0);
      unsigned char* dst = img + row * 128;
      int gr = m0 + row + (lane >> 3);
      opaque_v(gr);  // (keeps the row addresses from being hoisted out of the K-loop into registers the accumulators need)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(abase + (long)min(gr + 8 * j, g.M - 1) * lda + ko), (lptr_t)(dst + j * 1024), 16, 0, 0);
    } else {
      int row = rb0 + (r == 2 ? 32 : 0);
      unsigned char* dst = img + P4_OP + row * 128;
      int gr = n0 + row + (lane >> 3);
      opaque_v(gr);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(bbase + (long)min(gr + 8 * j, g.N - 1) * ldb + ko), (lptr_t)(dst + j * 1024), 16, 0, 0);
    }
  };
  // in front of barrier `beta` (0-based count of the loop's barriers): own pieces of every region up to beta + 2 have
  // landed; regions up to min(beta + 6, nr - 1) have been issued (2 DMA instructions per wave and region)
  auto wait_dma = [&](int beta) {
    const int ahead = min(beta + 6, nr - 1) - (beta + 2);
    if (ABL == 6) S2ST_VMCNT(12);
    else if (ahead >= 4) S2ST_VMCNT(8);
    else if (ahead == 3) S2ST_VMCNT(6);
    else if (ahead == 2) S2ST_VMCNT(4);
    else if (ahead == 1) S2ST_VMCNT(2);
    else S2ST_VMCNT(0);
  };
  // the last K-tile of a product with a K tail, before its first read: drain, publish, clear k >= K, publish
  auto clean_last = [&]() {
    S2ST_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    unsigned char* img = smem + ((nt - 1) & 1) * P4_IMG;
    const int kv = kend - (kbeg + (nt - 1) * BK);
    DA::sanitize(img, kv, tid);
    DB::sanitize(img + P4_OP, kv, tid);
    __syncthreads();
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment of 16 tile rows from row rt (rt % 16 == 0), k-half s: lane l holds row l & 15, k = 32 s + 8 (l >> 4) + j, at
  // chunk (4 s + (l >> 4)) ^ (row & 7) of its 128-byte row: s flips bit 6 of the lane's byte offset
  const int f0 = (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
  const int aoff = wm * WM * 128 + f0, boff = P4_OP + wn * WN * 128 + f0;
  bf16x8 a[2][4], bn0[2][2], bn1[2][2];
  // (reads by hand, s2st_asm.h: they stay in flight across MFMA groups and barriers; every use of a set is ordered behind
  // the counted wait that retires it by P4_USE_*; sched_barrier pins the instruction order)
#define P4_READ_A(S, IMG, R0)                                                                                            \
  a[S][0] = lds_read_b128_raw<((R0) + 0) * 128>((IMG) + (aoff ^ ((S) << 6)));                                           \
  a[S][1] = lds_read_b128_raw<((R0) + 16) * 128>((IMG) + (aoff ^ ((S) << 6)));                                          \
  a[S][2] = lds_read_b128_raw<((R0) + 32) * 128>((IMG) + (aoff ^ ((S) << 6)));                                          \
  a[S][3] = lds_read_b128_raw<((R0) + 48) * 128>((IMG) + (aoff ^ ((S) << 6)));
#define P4_READ_B(D, S, IMG, C0)                                                                                         \
  D[S][0] = lds_read_b128_raw<((C0) + 0) * 128>((IMG) + (boff ^ ((S) << 6)));                                           \
  D[S][1] = lds_read_b128_raw<((C0) + 16) * 128>((IMG) + (boff ^ ((S) << 6)));
#define P4_USE_A(S) lds_raw_fence(a[S][0]); lds_raw_fence(a[S][1]); lds_raw_fence(a[S][2]); lds_raw_fence(a[S][3]);
#define P4_USE_B(D, S) lds_raw_fence(D[S][0]); lds_raw_fence(D[S][1]);
#define P4_MFMA8(B_, S, I0, J0)                                                                                          \
  if (ABL == 5) __builtin_amdgcn_s_setprio(1);                                                                           \
  if (ABL != 4) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)              \
      acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B_[S][j], a[S][i], acc[(I0) + i][(J0) + j], 0, 0, 0); \
  if (ABL == 5) __builtin_amdgcn_s_setprio(0);
#define P4_FENCE() __builtin_amdgcn_sched_barrier(0)
  // counted LDS waits (reads return in order): N = the reads issued AFTER the ones the next MFMA group consumes
#define P4_LGKM(N) __builtin_amdgcn_s_waitcnt(0xc07f | ((N) << 8))
  // quadrant Q (1..4) of K-tile t opens: barrier 4 t + Q - 1
#define P4_OPEN(Q)                                                                                                       \
  wait_dma(4 * t + (Q) - 1);                                                                                             \
  if (ABL != 2 && !(ABL == 7 && ((Q) & 1) == 0)) __builtin_amdgcn_s_barrier();                                                                                          \
  P4_FENCE();
  // ... and, behind its first 8 MFMAs, re-fills the region the previous quadrant released: region 4 t + Q + 6 of the order
#define P4_FILL(Q)                                                                                                       \
  if (ABL != 1 && t + ((Q) + 6) / 4 < nt) issue(t + ((Q) + 6) / 4, ((Q) + 6) & 3);                                                   \
  P4_FENCE();

  {  // prologue: seven regions on their way; the first K-tile's A_m0 and B_n0 published; their fragments requested
    for (int n = 0; n < 7 && n < nr; ++n) {
      switch (n & 3) {
        case 0: issue(n >> 2, 0); break;
        case 1: issue(n >> 2, 1); break;
        case 2: issue(n >> 2, 2); break;
        default: issue(n >> 2, 3); break;
      }
    }
    if (tail && nt == 1) clean_last();
    else {
      if (nr >= 7) S2ST_VMCNT(10);  // regions 0 and 1 of the 7 issued
      else S2ST_VMCNT(4);           // (one K-tile: regions 0 and 1 of 4)
      __builtin_amdgcn_s_barrier();
    }
    P4_READ_A(0, smem, 0) P4_READ_B(bn0, 0, smem, 0)
    P4_READ_A(1, smem, 0) P4_READ_B(bn0, 1, smem, 0)
    P4_FENCE();
  }
  for (int t = 0; t < nt; ++t) {
    const unsigned char* cur = smem + (t & 1) * P4_IMG;
    const unsigned char* nxt = smem + ((t + 1) & 1) * P4_IMG;
    const bool more = t + 1 < nt;  // (wave-uniform, the same in every wave)
    // ---- quadrant (m0, n0); B_n1 is read ---------------------------------------------------------------------------
    P4_OPEN(1)
    P4_LGKM(6);
    P4_USE_A(0) P4_USE_B(bn0, 0)
    P4_MFMA8(bn0, 0, 0, 0)
    P4_FENCE();
    P4_FILL(1)
    if (ABL != 3) { P4_READ_B(bn1, 0, cur, 32) }
    P4_FENCE();
    P4_LGKM(2);
    P4_USE_A(1) P4_USE_B(bn0, 1)
    P4_MFMA8(bn0, 1, 0, 0)
    P4_FENCE();
    if (ABL != 3) { P4_READ_B(bn1, 1, cur, 32) }
    P4_FENCE();
    // ---- quadrant (m0, n1); A_m1 is read ---------------------------------------------------------------------------
    P4_OPEN(2)
    P4_LGKM(2);
    P4_USE_B(bn1, 0)
    P4_MFMA8(bn1, 0, 0, 2)
    P4_FENCE();
    P4_FILL(2)
    if (ABL != 3) { P4_READ_A(0, cur, 64) }
    P4_FENCE();
    P4_LGKM(4);
    P4_USE_B(bn1, 1)
    P4_MFMA8(bn1, 1, 0, 2)
    P4_FENCE();
    if (ABL != 3) { P4_READ_A(1, cur, 64) }
    P4_FENCE();
    // ---- quadrant (m1, n1) ------------------------------------------------------------------------------------------
    P4_OPEN(3)
    P4_LGKM(4);
    P4_USE_A(0)
    P4_MFMA8(bn1, 0, 4, 2)
    P4_FENCE();
    P4_FILL(3)
    P4_LGKM(0);
    P4_USE_A(1)
    P4_MFMA8(bn1, 1, 4, 2)
    P4_FENCE();
    // ---- quadrant (m1, n0); the next K-tile's A_m0 and B_n0 are read -----------------------------------------------
    P4_OPEN(4)
    if (tail && t == nt - 2) clean_last();
    P4_MFMA8(bn0, 0, 4, 0)
    P4_FENCE();
    P4_FILL(4)
    if (ABL != 3 && more) { P4_READ_A(0, nxt, 0) P4_READ_B(bn0, 0, nxt, 0) }
    P4_FENCE();
    P4_MFMA8(bn0, 1, 4, 0)
    P4_FENCE();
    if (ABL != 3 && more) { P4_READ_A(1, nxt, 0) P4_READ_B(bn0, 1, nxt, 0) }
    P4_FENCE();
  }
#undef P4_OPEN
#undef P4_FILL
#undef P4_LGKM
#undef P4_MFMA8
#undef P4_READ_A
#undef P4_READ_B
#undef P4_USE_A
#undef P4_USE_B
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

template <int ABL>
__global__ __launch_bounds__(64 * P4_NW) void gemm_bf16_p4_kernel(GemmArgs g) {
  gemm_p4_tile<ABL>(g, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
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
template <int ABL>
static int p4_launch_abl(const GemmArgs& g, dim3 grid, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_p4_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            P4_LDS) != hipSuccess)
      return -1;
    configured = true;
  }
  s2st_launch("gemm_bf16_p4_kernel<256, 256>", p4_flops(g), p4_min_bytes(g), gemm_bf16_p4_kernel<ABL>, grid, dim3(64 * P4_NW), P4_LDS,
              st, g);
  return 0;
}
int s2st_gemm_bf16_p4(const GemmArgs& g, dim3 grid, hipStream_t st) {
  if (!g.A.kmajor || !g.B.kmajor) return S2ST_ERR_ARG;
  switch (s2st_env_int("S2ST_P4_ABL", 0)) {  // DEV ONLY
    case 1: return p4_launch_abl<1>(g, grid, st);
    case 2: return p4_launch_abl<2>(g, grid, st);
    case 3: return p4_launch_abl<3>(g, grid, st);
    case 4: return p4_launch_abl<4>(g, grid, st);
    case 5: return p4_launch_abl<5>(g, grid, st);
    case 6: return p4_launch_abl<6>(g, grid, st);
    case 7: return p4_launch_abl<7>(g, grid, st);
    default: return p4_launch_abl<0>(g, grid, st);
  }
}

int s2st_gemm_bf16_p4_preload(hipStream_t st) {
  GemmArgs g{};
  g.A.dtype = g.B.dtype = S2ST_BF16;
  g.A.kmajor = g.B.kmajor = 1;
  g.splitk = 1; g.zdiv = 1; g.tiles_n = 1; g.batch = 1; g.kchunk = BK;
  const int rc = s2st_gemm_bf16_p4(g, dim3(1), st);
  return rc || hipGetLastError() != hipSuccess ? -1 : 0;
}
