// bf16 GEMM, 256 x 256 tile in four phases per K-tile (round 5): C(m, n) = epi(alpha * sum_k A(m, k) B(n, k)), both
// operands K-contiguous (the forward linear layers; data gradients against the pre-transposed weight copies; HuBERT).
//
// Why a third form.  The 128-row ring kernels need 64 B/clk/CU of operand fill to keep the MFMAs busy (32 KB per 64-deep
// K-step for 2.1 MFLOP = 512 MFMA clocks) and a CU takes in ~34 B/clk (MI355X_MICROARCH.md, gather / ldsdma-fill rows;
// measured here: 950 clocks per K-step): they are fill-bound at about half the MFMA rate, whatever their pipelining.  A
// 256 x 256 tile moves 64 KB per K-step for 8.4 MFLOP = 2048 MFMA clocks, i.e. needs 32 B/clk/CU -- the first tile shape
// that CAN be matrix-bound on this chip (cdna_hip_programming.md section 5, "The 256^2 8-phase template").  It is picked
// where a product has enough 256 x 256 tiles to occupy the chip: HuBERT's projections (M = 9.6 k ... 307 k rows), and the
// N >= 1536 products of the training step at M = 3.4 - 4.6 k.
//
// Structure (8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 MFMA tiles of 16 x 16, 128 accumulator registers):
//   * two K-tile buffers of 64 KB (A 256 x 64, B 256 x 64, the K-contiguous swizzled images of gemm_bf16_tile.h), filled
//     by LDS-DMA (8 wave-instructions of 1 KB per wave and K-tile);
//   * a K-tile is multiplied as four QUADRANTS of the wave tile (64 x 32 each, 16 MFMAs): (m-half 0, n-half 0) ->
//     (0, 1) -> (1, 1) -> (1, 0), so that consecutive quadrants share one operand's fragments and replace the other's:
//     live fragments are 32 (an A half) + 16 (a B half) registers instead of 96 for the whole K-tile -- what lets the
//     accumulators, the fragments and the addresses fit the 256 registers of a wave at two waves per SIMD;
//   * fragment reads of the NEXT quadrant are issued between the MFMAs of the current one, k-half by k-half (the
//     registers of a k-half are free as soon as its MFMAs have issued; the first n-half is read twice per K-tile): the
//     only reads not covered by MFMAs of the same wave are the twelve right behind the K-tile's barrier;
//   * the DMA of K-tile t + 1 is issued during the first two quadrants of K-tile t (its buffer was released by the
//     barrier that opened K-tile t) and waited for with vmcnt(0) right before the barrier that opens K-tile t + 1: one
//     barrier per K-tile.  (The guide's template keeps three half-tiles in flight across its barriers with counted waits
//     and staggers the two wave rows by a barrier; this form keeps the synchronisation provable at a glance -- one wait,
//     one barrier, no read of a buffer while any DMA into it is outstanding -- and takes the rest as a later step.)
// Tile images, swizzles, the K-tail rule (clamped loads, zeroed in LDS) and the epilogues are those of the ring kernels.
// Replaces F.linear of fairseq/modules/transformer_layer.py:140-162, multihead_attention.py:170-192 and
// fairseq/models/wav2vec/wav2vec2.py:736-814, 915-1016 (HuBERT projections / convolutions as GEMMs) in fast mode.
#include "gemm_bf16_tile.h"

namespace {

constexpr int P4_BM = 256, P4_BN = 256, P4_NW = 8, P4_WGN = 4;
constexpr int P4_LDS = 2 * (P4_BM + P4_BN) * 128;  // 128 KB

__device__ __forceinline__ void gemm_p4_tile(const GemmArgs& g, int id, const int nwg, const int by) {
  constexpr int BM = P4_BM, BN = P4_BN, NW = P4_NW, WGN = P4_WGN;
  constexpr int WM = BM / 2, WN = BN / WGN, TM = WM / 16, TN = WN / 16;  // 128 x 64: 8 x 4 MFMA tiles
  static_assert(TM == 8 && TN == 4, "wave tile");
  typedef Dma<true, BM, NW> DA;  // (image geometry and the K-tail rule; the loads themselves are issued below)
  typedef Dma<true, BN, NW> DB;
  typedef Stage<true, BM, true> LA;
  typedef Stage<true, BN, true> LB;
  constexpr int A_BYTES = DA::BYTES, STAGE = DA::BYTES + DB::BYTES;
  static_assert(DA::NI == 4 && DB::NI == 4, "DMA pieces per wave and operand");
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

  // LDS-DMA source addresses are formed on the fly (a handful of VALU operations per 1 KB piece against 64 MFMAs per
  // K-tile) from three registers -- the first row this lane fetches of either operand and its swizzled chunk -- instead
  // of eight 64-bit pointers: the registers the accumulators need.  Plain row strides only (no conv window addressing:
  // p4_pick leaves those products to the ring kernels).  Piece j of an operand = rows [32 wave + 8 j, + 8) of the tile,
  // lane l -> row l >> 3, 16-byte slot l & 7, source chunk (l & 7) ^ (l >> 3) (the image's swizzle, gemm_bf16_tile.h).
  const int chunk8 = 8 * ((lane & 7) ^ (lane >> 3));
  const int rowa = m0 + wave * 32 + (lane >> 3), rowb = n0 + wave * 32 + (lane >> 3);
  const long lda = g.A.sp.ld, ldb = g.B.sp.ld;
  const int kpad = ((g.K + 7) & ~7) - 8;  // (loads stay inside the padded row: the K tail is zeroed in LDS)
  // (the opaque copy of the row keeps the compiler from hoisting the eight loop-invariant row addresses out of the K-loop,
  // i.e. from turning the three registers back into sixteen)
  auto issue_a = [&](unsigned char* img, int kt) {
    const int ko = min(kt + chunk8, kpad);
    int r = rowa;
    opaque_v(r);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(abase + (long)min(r + 8 * j, g.M - 1) * lda + ko),
                                       (lptr_t)(img + (wave * 32 + j * 8) * 128), 16, 0, 0);
  };
  auto issue_b = [&](unsigned char* img, int kt) {
    const int ko = min(kt + chunk8, kpad);
    int r = rowb;
    opaque_v(r);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(bbase + (long)min(r + 8 * j, g.N - 1) * ldb + ko),
                                       (lptr_t)(img + A_BYTES + (wave * 32 + j * 8) * 128), 16, 0, 0);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nt > 0) {
    issue_a(smem, kbeg);
    issue_b(smem, kbeg);
  }
  const int arow = wm * WM, bcol = wn * WN;
  for (int t = 0; t < nt; ++t) {
    // K-tile t has landed (this wave's pieces: vmcnt(0) -- nothing younger is in flight; everyone's: the barrier), and
    // every wave is done reading the other buffer (K-tile t - 1), which the DMA of K-tile t + 1 is about to overwrite
    S2ST_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    unsigned char* cur = smem + (t & 1) * STAGE;
    unsigned char* nxt = smem + ((t + 1) & 1) * STAGE;
    const bool more = t + 1 < nt;  // (wave-uniform, the same in every wave)
    const int kn = kbeg + (t + 1) * BK;
    if (!more && kend - (kbeg + t * BK) < BK) {  // K tail: zero the invalid k of this K-tile
      const int kv = kend - (kbeg + t * BK);
      DA::sanitize(cur, kv, tid);
      DB::sanitize(cur + A_BYTES, kv, tid);
      __syncthreads();
    }
    const unsigned char* aimg = cur;
    const unsigned char* bimg = cur + A_BYTES;
    // ONE set of fragment registers per operand: a[k-half][4 m-tiles] (32 registers), b[k-half][2 n-tiles] (16).  A quadrant
    // multiplies k-half 0 then k-half 1 (8 MFMAs each, all on different accumulators); the registers of a k-half are
    // re-filled with the NEXT quadrant's fragments right behind that k-half's MFMAs, so every read but the first twelve
    // of a K-tile has at least 8 MFMAs (128 matrix-pipe clocks) in front of its first use.  sched_barrier pins the order:
    // left alone, the compiler hoists the reads, runs out of registers and spills accumulators inside the loop.
    bf16x8 a[2][4], b[2][2];
#define P4_MFMA8(S, I0, J0)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                            \
      acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[S][j], a[S][i], acc[(I0) + i][(J0) + j], 0, 0, 0);
#define P4_READ_A(S, R0) _Pragma("unroll") for (int i = 0; i < 4; ++i) a[S][i] = LA::frag(aimg, arow + (R0) + i * 16, S, lane);
#define P4_READ_B(S, C0) _Pragma("unroll") for (int j = 0; j < 2; ++j) b[S][j] = LB::frag(bimg, bcol + (C0) + j * 16, S, lane);
#define P4_FENCE() __builtin_amdgcn_sched_barrier(0)
    // counted LDS waits (reads return in order): N = the reads issued AFTER the ones the next MFMA group consumes.  hipcc
    // waits lgkmcnt(0) at these points -- i.e. also for the reads it has just issued for the group after next, whose latency
    // the 8 MFMAs in between are there to cover
#define P4_LGKM(N) __builtin_amdgcn_s_waitcnt(0xc07f | ((N) << 8))
    // ---- quadrant (m-half 0, n-half 0) + the A half of the next K-tile's DMA -----------------------------------------
    P4_READ_B(0, 0) P4_READ_A(0, 0) P4_READ_B(1, 0) P4_READ_A(1, 0)
    if (more) issue_a(nxt, kn);
    P4_FENCE();
    P4_LGKM(6);
    P4_MFMA8(0, 0, 0)
    P4_FENCE();
    P4_READ_B(0, 32)  // (n-half 1, k-half 0) into the registers k-half 0 just released
    P4_FENCE();
    P4_LGKM(2);
    P4_MFMA8(1, 0, 0)
    P4_FENCE();
    P4_READ_B(1, 32)
    // ---- quadrant (0, 1) + the B half of the next K-tile's DMA -----------------------------------------------------
    if (more) issue_b(nxt, kn);
    P4_FENCE();
    P4_LGKM(2);
    P4_MFMA8(0, 0, 2)
    P4_FENCE();
    P4_READ_A(0, 64)  // (m-half 1, k-half 0)
    P4_FENCE();
    P4_LGKM(4);
    P4_MFMA8(1, 0, 2)
    P4_FENCE();
    P4_READ_A(1, 64)
    P4_FENCE();
    // ---- quadrant (1, 1) ---------------------------------------------------------------------------------------------
    P4_LGKM(4);
    P4_MFMA8(0, 4, 2)
    P4_FENCE();
    P4_READ_B(0, 0)  // back to n-half 0
    P4_FENCE();
    P4_LGKM(2);
    P4_MFMA8(1, 4, 2)
    P4_FENCE();
    P4_READ_B(1, 0)
    P4_FENCE();
    // ---- quadrant (1, 0) ---------------------------------------------------------------------------------------------
    P4_LGKM(2);
    P4_MFMA8(0, 4, 0)
    P4_FENCE();
    P4_LGKM(0);
    P4_MFMA8(1, 4, 0)
    P4_FENCE();
#undef P4_LGKM
#undef P4_MFMA8
#undef P4_READ_A
#undef P4_READ_B
#undef P4_FENCE
  }
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
