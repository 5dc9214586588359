// bf16-operand GEMM for gfx950:  C(m,n) = epi(alpha * sum_k A(m,k) * B(n,k)),  A and B bf16 in HBM,
// fp32 accumulation on v_mfma_f32_16x16x32_bf16, fp32 and/or bf16 result.
//
// This is the fast path of s2st_gemm (gemm.hip keeps the fp32-operand kernel used by the bf16x3
// "precise" mode): the training engine stores a bf16 copy of every tensor that is a GEMM operand
// (weights, layer-norm outputs, attention probabilities, activation gradients), so the values the
// matrix cores see are bit-identical to what the fp32 path produces by converting on the fly --
// at half the HBM/L2 bytes and with no conversion VALU in the loop.
//
// Structure: 256 threads = 4 waves (2 x 2), block tile BM x BN in {128x128, 128x64, 64x64},
// BK = 64 (two MFMA k-steps per barrier).  Tiles are staged through registers (16-byte global
// loads of tile t+1 are in flight during the MFMAs of tile t) into a double-buffered LDS image:
//   K-contiguous operand   [row][64 k]    128-byte rows, 16-byte chunk c stored at slot c ^ (row & 7)
//                                         -> ds_read_b128 fragment reads are bank-conflict free;
//   rows-contiguous operand [k][ROWS]     natural layout (weight-gradient / P*V / data-gradient
//                                         forms), 32-byte pair c stored at c ^ f(k) so that the 8 k
//                                         rows a half-wave reads with ds_read_b64_tr_b16 hit
//                                         distinct banks.
// The MFMA is issued with the operands swapped (D = B_frag x A_frag = C^T tile) so that each lane
// ends up with 4 consecutive n of one row m: the epilogue (alpha, bias, ReLU, dropout, residual,
// accumulate) runs on float4 and stores 16 bytes (fp32) / 8 bytes (bf16 copy) per lane.
// Workgroup ids are remapped so that the blocks that share an XCD (id % 8) own a contiguous run
// of tiles (same A rows -> hits in that XCD's L2).
#include "gemm_bf16_tile.h"

namespace {

template <int BM, int BN, bool AKM, bool BKM, bool VEC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
  typedef Stage<AKM, BM, VEC> LA;
  typedef Stage<BKM, BN, VEC> LB;
  constexpr int A_BYTES = LA::BYTES, B_BYTES = LB::BYTES, STAGE = A_BYTES + B_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: ids that share an XCD (id % 8) get a contiguous run of tiles
  const int nwg = gridDim.x;
  int id = blockIdx.x;
  {
    const int x = id & 7, q = nwg >> 3, r = nwg & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
  }
  const int tile_m = id / g.tiles_n, tile_n = id - tile_m * g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int zb = blockIdx.y / g.splitk, ks = blockIdx.y - zb * g.splitk;
  const int zq = zb / g.zdiv, zr = zb - zq * g.zdiv;
  const bf16_t* abase = reinterpret_cast<const bf16_t*>(g.A.p) + zq * g.A.zo + zr * g.A.zi;
  const bf16_t* bbase = reinterpret_cast<const bf16_t*>(g.B.p) + zq * g.B.zo + zr * g.B.zi;
  const int kbeg = ks * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nt = (kend - kbeg + BK - 1) / BK;

  LA la;
  LB lb;
  la.init(g.A, m0, g.M, tid);
  lb.init(g.B, n0, g.N, tid);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nt > 0) {
    la.load(g.A, abase, g.M, kbeg, kend);
    lb.load(g.B, bbase, g.N, kbeg, kend);
    la.store(smem);
    lb.store(smem + A_BYTES);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const unsigned char* cur = smem + (t & 1) * STAGE;
    unsigned char* nxt = smem + ((t + 1) & 1) * STAGE;
    const bool more = t + 1 < nt;
    if (more) {
      la.load(g.A, abase, g.M, kbeg + (t + 1) * BK, kend);
      lb.load(g.B, bbase, g.N, kbeg + (t + 1) * BK, kend);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = LA::frag(cur, wm * WM + i * 16, s, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = LB::frag(cur + A_BYTES, wn * WN + j * 16, s, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // swapped operands: D[n][m] -> lane: m = lane & 15, n = 4 (lane >> 4) + r
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (more) {
      la.store(nxt);
      lb.store(nxt + A_BYTES);
    }
    __syncthreads();
  }

  gemm_epilogue<BM, BN, 2>(g, acc, m0, n0, wm, wn, lane, zb, ks, zq, zr);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA ring kernel (aligned operands): the same tile math, but the tiles are moved HBM/L2 -> LDS by
// global_load_lds_dwordx4 (no VGPR staging, no ds_write) into a ring of NS stages, NS-1 K-steps ahead
// of the MFMAs.  A wave-instruction writes 1 KiB of LDS linearly (lane l -> base + 16 l), so the
// swizzles of the LDS images above are applied to the per-lane SOURCE address instead:
//   K-contiguous image : a wave-instruction covers 8 rows x 128 B; lane l = (row l>>3, slot l&7) fetches
//                        chunk (l&7) ^ (l>>3) of its row;
//   rows-contiguous    : a wave-instruction covers 1 KiB of k rows; slot p of k row kr fetches chunk
//                        (((p>>1) ^ kperm(kr)) << 1) | (p&1).
// One raw s_barrier per K-step: [counted vmcnt: stage t landed] -> barrier (everyone's stage-t DMA is
// visible, everyone is done reading stage t-1) -> issue stage t+NS-1 into the slot of t-1 -> MFMAs of
// stage t.  A K tail (kend - kt < 64) is loaded from clamped addresses and zeroed in LDS.

// One tile of one problem: `id` = the workgroup's index among the `nwg` tiles of the problem, `by` = its batch /
// split-K coordinate (the launch's blockIdx.y; 0 for grouped launches)
template <int BM, int BN, bool AKM, bool BKM, int NS, int NW, bool IL>
__device__ __forceinline__ void gemm_bf16_dma_tile(const GemmArgs& g, int id, const int nwg, const int by) {
  constexpr int WGN = NW / 2;  // waves: 2 (M) x WGN (N)
  constexpr int WM = BM / 2, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  typedef Dma<AKM, BM, NW> DA;
  typedef Dma<BKM, BN, NW> DB;
  typedef Stage<AKM, BM, true> LA;  // fragment readers (same images)
  typedef Stage<BKM, BN, true> LB;
  constexpr int A_BYTES = DA::BYTES, B_BYTES = DB::BYTES, STAGE = A_BYTES + B_BYTES;
  constexpr int PER_STAGE = DA::NI + DB::NI;  // DMA instructions per wave per stage
  HIP_DYNAMIC_SHARED(unsigned char, smem)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  if (nwg > 0) {  // (nwg == 0: the caller already placed the tiles)
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

#ifdef S2ST_GEMM_STAMP  // tools/gemm_stamp.sh: a private build that writes per-workgroup cycle stamps into g.ws
  long stamp[6];
  stamp[0] = clock64();
#endif
  DA da;
  DB db;
  da.init(g.A, abase, m0, g.M, wave, lane);
  db.init(g.B, bbase, n0, g.N, wave, lane);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: stages 0 .. NS-2
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nt) {
      da.issue(smem + s * STAGE, kbeg + s * BK, g.K, wave);
      db.issue(smem + s * STAGE + A_BYTES, kbeg + s * BK, g.K, wave);
    }

#ifdef S2ST_GEMM_STAMP
  stamp[1] = clock64();
  S2ST_VMCNT((NS - 2) * PER_STAGE);
  __builtin_amdgcn_s_barrier();
  stamp[2] = clock64();
#endif
  int t0 = 0;
  if constexpr (IL) {
    // Steady state as ONE basic block (no refill / K-tail conditions: those steps are left to the loop below) so that
    // the scheduler can be told to spread the K-step's DMA pieces and fragment reads between its MFMAs: issued in
    // three separate bursts behind the barrier, every wave of the CU is in its DMA phase (one texture-address path per
    // CU, ~100 cycles of issue per 1 KB piece), then its LDS phase, then its MFMA phase, and the three do not overlap.
    if constexpr (AKM && BKM) {
      for (; t0 + NS - 1 < nt; ++t0) {
        S2ST_VMCNT((NS - 2) * PER_STAGE);
        __builtin_amdgcn_s_barrier();
        unsigned char* cur = smem + (t0 % NS) * STAGE;
        unsigned char* nxt = smem + ((t0 + NS - 1) % NS) * STAGE;
        bf16x8 af[2][TM], bf[2][TN];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[s][j] = LB::frag(cur + A_BYTES, wn * WN + j * 16, s, lane);
#pragma unroll
          for (int i = 0; i < TM; ++i) af[s][i] = LA::frag(cur, wm * WM + i * 16, s, lane);
        }
        da.issue(nxt, kbeg + (t0 + NS - 1) * BK, g.K, wave);
        db.issue(nxt + A_BYTES, kbeg + (t0 + NS - 1) * BK, g.K, wave);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
        // the order the scheduler has to produce: the fragments of the first half K-step; under its MFMAs the fragment
        // reads of the second half; under the second half's MFMAs the DMA pieces of the stage NS - 1 steps ahead
        constexpr int RB = LB::READS_PER_FRAG, RA = LA::READS_PER_FRAG;
        constexpr int HALF = TN * RB + TM * RA;  // LDS reads per half K-step
        sched_group<0x100, HALF>();
        sched_rows<0, TM, TN, HALF, 0x100>();
        sched_rows<0, TM, TN, PER_STAGE, 0x010>();
      }
    } else {
      // a rows-contiguous operand: its transposed reads are hand-issued (s2st_asm.h), which the scheduler's groups cannot
      // place -- the same order written out, pinned by sched_barrier(0): first half's fragments | per accumulator row of
      // the first half: its MFMAs, then a share of the second half's reads | per row of the second half: its MFMAs, then
      // one DMA piece of the stage NS - 1 steps ahead
      for (; t0 + NS - 1 < nt; ++t0) {
        S2ST_VMCNT((NS - 2) * PER_STAGE);
        __builtin_amdgcn_s_barrier();
        unsigned char* cur = smem + (t0 % NS) * STAGE;
        unsigned char* nxt = smem + ((t0 + NS - 1) % NS) * STAGE;
        const int kn = kbeg + (t0 + NS - 1) * BK;
        typename LA::Raw ar[2][TM];
        typename LB::Raw br[2][TN];
        bf16x8 af[2][TM], bf[2][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) br[0][j] = LB::raw(cur + A_BYTES, wn * WN + j * 16, 0, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i) ar[0][i] = LA::raw(cur, wm * WM + i * 16, 0, lane);
        lds_raw_wait();
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = LB::done(br[0][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = LA::done(ar[0][i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[0][i], acc[i][j], 0, 0, 0);
          if (i == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) br[1][j] = LB::raw(cur + A_BYTES, wn * WN + j * 16, 1, lane);
          }
          ar[1][i] = LA::raw(cur, wm * WM + i * 16, 1, lane);
          __builtin_amdgcn_sched_barrier(0);
        }
        lds_raw_wait();
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[1][j] = LB::done(br[1][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[1][i] = LA::done(ar[1][i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1][j], af[1][i], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int q = (PER_STAGE * i) / TM; q < (PER_STAGE * (i + 1)) / TM; ++q) {
            if (q < DA::NI) da.issue_piece(nxt, kn, g.K, wave, q);
            else db.issue_piece(nxt + A_BYTES, kn, g.K, wave, q - DA::NI);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  for (int t = t0; t < nt; ++t) {
    // my DMA of stage t has landed once at most min(NS-2, nt-1-t) later stages are outstanding
    const int ahead = nt - 1 - t;
    if (ahead >= NS - 2) S2ST_VMCNT((NS - 2) * PER_STAGE);
    else if (NS > 3 && ahead == 1) S2ST_VMCNT(PER_STAGE);
    else S2ST_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    unsigned char* cur = smem + (t % NS) * STAGE;
    if (t + NS - 1 < nt) {
      unsigned char* nxt = smem + ((t + NS - 1) % NS) * STAGE;
      da.issue(nxt, kbeg + (t + NS - 1) * BK, g.K, wave);
      db.issue(nxt + A_BYTES, kbeg + (t + NS - 1) * BK, g.K, wave);
    }
    if (t == nt - 1 && kend - (kbeg + t * BK) < BK) {  // K tail: zero the invalid k of this stage
      const int kv = kend - (kbeg + t * BK);
      DA::sanitize(cur, kv, tid);
      DB::sanitize(cur + A_BYTES, kv, tid);
      __syncthreads();
    }
    // all fragment reads of the K-step are issued up front (B first: every MFMA row needs all of it),
    // the MFMAs then consume them in issue order behind counted lgkmcnt waits: LDS latency is paid once
    // per K-step instead of once per read group
    bf16x8 af[2][TM], bf[2][TN];
    {
      typename LA::Raw ar[2][TM];
      typename LB::Raw br[2][TN];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < TN; ++j) br[s][j] = LB::raw(cur + A_BYTES, wn * WN + j * 16, s, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i) ar[s][i] = LA::raw(cur, wm * WM + i * 16, s, lane);
      }
      if (!AKM || !BKM) lds_raw_wait();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[s][j] = LB::done(br[s][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[s][i] = LA::done(ar[s][i]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
  }
#ifdef S2ST_GEMM_STAMP
  stamp[3] = clock64();
#endif
  gemm_epilogue<BM, BN, WGN, IL>(g, acc, m0, n0, wm, wn, lane, zb, ks, zq, zr);  // (the production instantiations carry the fast forms)
#ifdef S2ST_GEMM_STAMP
  stamp[4] = clock64();
  S2ST_VMCNT(0);
  stamp[5] = clock64();
  if (g.ws && tid == 0 && g.splitk == 1) {
    long* o = reinterpret_cast<long*>(g.ws) + (long)blockIdx.x * 8;
    for (int i = 0; i < 6; ++i) o[i] = stamp[i];
  }
#endif
}

template <int BM, int BN, bool AKM, bool BKM, int NS, int NW, bool IL = false>
__global__ __launch_bounds__(64 * NW) void gemm_bf16_dma_kernel(GemmArgs g) {
  gemm_bf16_dma_tile<BM, BN, AKM, BKM, NS, NW, IL>(g, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

// Grouped one-shot form: the concatenated tile list of up to S2ST_GROUP_MAX problems (batch 1, K unsplit), one
// workgroup per tile -- the plain K-loop above (no tile walk, no cursors: straight-line steady state) for a layer's
// weight-gradient products, whose 72 K-steps amortise the prologue by themselves.
template <int BM, int BN, bool AKM, bool BKM, int NS, int NW, bool IL = false>
__global__ __launch_bounds__(64 * NW) void gemm_bf16_dma_group_kernel(GemmGroup grp) {
  int pi = 0;
  int t = blockIdx.x;
  if (grp.xcd_global) {
    // The workgroups of one XCD (blockIdx % 8) take ONE contiguous run of the concatenated tile list: ~G / 8 tiles of one
    // or two products, i.e. a few whole tile rows -- they share far fewer operand panels in that XCD's L2 than one
    // eighth of EVERY product's tiles (the per-product order below), which is what re-fetched every panel ~2x per launch.
    const int G = grp.tile0[grp.n];
    const int x = t & 7, q = G >> 3, r = G & 7;
    t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3);
  }
  while (pi + 1 < grp.n && t >= grp.tile0[pi + 1]) ++pi;
  gemm_bf16_dma_tile<BM, BN, AKM, BKM, NS, NW, IL>(grp.g[pi], t - grp.tile0[pi],
                                                    grp.xcd_global ? 0 : grp.tile0[pi + 1] - grp.tile0[pi], 0);
}

// as-launched work of one GEMM (profiling records): FLOPs, and the bytes it has to move at least -- both bf16
// operands once, the result once per output copy, the old value / residual once when it is read
double gemm_flops(const GemmArgs& g) { return 2.0 * g.M * g.N * (double)g.K * g.batch; }
double gemm_min_bytes(const GemmArgs& g) {
  const double mn = (double)g.M * g.N * g.batch;
  return 2.0 * g.batch * ((double)g.M * g.K + (double)g.N * g.K) + mn * ((g.C.p ? 4 : 0) + (g.C.h ? 2 : 0)) +
         mn * 4 * ((g.ep.accumulate ? 1 : 0) + (g.ep.resid ? 1 : 0));
}
// profiling tag of an instantiation, spelled like the kernel name in a rocprofv3 kernel trace
template <int BM, int BN, bool AKM, bool BKM, int NS, int NW, bool IL = false>
const char* dma_tag() {
  static char buf[96];
  if (!buf[0])
    snprintf(buf, sizeof buf, "gemm_bf16_dma_kernel<%d, %d, %s, %s, %d, %d, %s>", BM, BN, AKM ? "true" : "false",
             BKM ? "true" : "false", NS, NW, IL ? "true" : "false");
  return buf;
}
template <int BM, int BN, bool AKM, bool BKM, bool VEC>
const char* staged_tag() {
  static char buf[80];
  if (!buf[0])
    snprintf(buf, sizeof buf, "gemm_bf16_kernel<%d, %d, %s, %s, %s>", BM, BN, AKM ? "true" : "false",
             BKM ? "true" : "false", VEC ? "true" : "false");
  return buf;
}

template <int BM, int BN, int NS, int NW, bool IL = false>
int launch_dma(const GemmArgs& g, dim3 grid, hipStream_t st) {
  constexpr int LDS = NS * (BM + BN) * 128;
  auto go = [&](auto kern, const char* tag) {
    static bool configured = false;  // one flag per instantiation (the lambda's operator() template)
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
        return -1;
      configured = true;
    }
    s2st_launch(tag, gemm_flops(g), gemm_min_bytes(g), kern, grid, dim3(64 * NW), LDS, st, g);
    return 0;
  };
  if (g.A.kmajor && g.B.kmajor) return go(gemm_bf16_dma_kernel<BM, BN, true, true, NS, NW, IL>, dma_tag<BM, BN, true, true, NS, NW, IL>());
  if (g.A.kmajor && !g.B.kmajor) return go(gemm_bf16_dma_kernel<BM, BN, true, false, NS, NW, IL>, dma_tag<BM, BN, true, false, NS, NW, IL>());
  if (!g.A.kmajor && g.B.kmajor) return go(gemm_bf16_dma_kernel<BM, BN, false, true, NS, NW, IL>, dma_tag<BM, BN, false, true, NS, NW, IL>());
  return go(gemm_bf16_dma_kernel<BM, BN, false, false, NS, NW, IL>, dma_tag<BM, BN, false, false, NS, NW, IL>());
}

template <int BM, int BN, int NS, int NW, bool IL = false>
int launch_dma_group(const GemmGroup& grp, hipStream_t st) {
  constexpr int LDS = NS * (BM + BN) * 128;
  double fl = 0, by = 0;
  for (int i = 0; i < grp.n; ++i) { fl += gemm_flops(grp.g[i]); by += gemm_min_bytes(grp.g[i]); }
  auto go = [&](auto kern, const char* tag) {
    static bool configured = false;
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
        return -1;
      configured = true;
    }
    s2st_launch(tag, fl, by, kern, dim3(grp.total < 1 ? 1 : grp.total), dim3(64 * NW), LDS, st, grp);
    return 0;
  };
  const bool akm = grp.g[0].A.kmajor != 0, bkm = grp.g[0].B.kmajor != 0;
  static char tags[4][96];
  auto tag = [&](int i) {
    if (!tags[i][0])
      snprintf(tags[i], sizeof tags[i], "gemm_bf16_dma_group_kernel<%d, %d, %s, %s, %d, %d, %s>", BM, BN, (i & 2) ? "true" : "false",
               (i & 1) ? "true" : "false", NS, NW, IL ? "true" : "false");
    return (const char*)tags[i];
  };
  if (akm && bkm) return go(gemm_bf16_dma_group_kernel<BM, BN, true, true, NS, NW, IL>, tag(3));
  if (akm && !bkm) return go(gemm_bf16_dma_group_kernel<BM, BN, true, false, NS, NW, IL>, tag(2));
  if (!akm && bkm) return go(gemm_bf16_dma_group_kernel<BM, BN, false, true, NS, NW, IL>, tag(1));
  return go(gemm_bf16_dma_group_kernel<BM, BN, false, false, NS, NW, IL>, tag(0));
}

// The persistent tile walk, its stream-K form and the 256 x 128 tile were built and measured in rounds 2 - 3 (DESIGN.md
// section 4 / 5: none of them is what the launcher picks for this step's products).  They are compiled only into
// -DS2ST_EXPERIMENTAL builds (tools/build_experimental.sh: the A/B tools, the emulator's test build), so that the product
// library carries -- and preloads -- only the instantiations its pickers can select.
#ifdef S2ST_EXPERIMENTAL
// ------------------------------------------------------------------------------------------------
// Persistent, grouped form of the ring kernel.  One workgroup per CU walks tiles t = id, id + G, id + 2G, ... of the
// concatenated tile list of up to S2ST_GROUP_MAX problems (same operand layouts, batch 1, no split-K), and the DMA
// ring runs ACROSS tile and problem boundaries: while a tile's last K-steps are multiplied -- and while its epilogue
// stores drain -- the first stages of the next tile are already landing.  The one-shot kernel above pays a full
// prologue (every CU bursting its first three stages at once: ~4 us) and an epilogue per round of tiles, which for
// the K = 512 ... 2048 products of a training step is most of a launch.  Grouping is what the weight-gradient GEMMs
// of a layer use: four launches (+ split-K slabs and their combine kernels) become one with K = tokens unsplit.
// vmcnt counts loads, LDS-DMA and stores together in issue order: right after an epilogue the stores are the
// YOUNGEST operations, so the counted wait "all but the (stages ahead) x PER_STAGE youngest" still covers the stage
// about to be read (it may wait for a few stores too: correct, slightly conservative).
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, bool AKM, bool BKM, int NS, int NW, bool SK = false>
__global__ __launch_bounds__(64 * NW) void gemm_bf16_dma_persistent_kernel(GemmGroup grp) {
  constexpr int WGN = NW / 2;
  constexpr int WM = BM / 2, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  typedef Dma<AKM, BM, NW> DA;
  typedef Dma<BKM, BN, NW> DB;
  typedef Stage<AKM, BM, true> LA;
  typedef Stage<BKM, BN, true> LB;
  constexpr int A_BYTES = DA::BYTES, B_BYTES = DB::BYTES, STAGE = A_BYTES + B_BYTES;
  constexpr int PER_STAGE = DA::NI + DB::NI;
  static_assert(NS >= 3 && NS <= 5, "ring depth");
  HIP_DYNAMIC_SHARED(unsigned char, smem)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  const int G = gridDim.x, total = grp.total;

  // ---- stream-K: this workgroup's share [s_begin, s_end) of the K-steps of all tiles ---------------------------
  // Logical index w = (XCD of the workgroup) * G/8 + ticket: workgroups of one XCD get adjacent ranges (their tiles share
  // operand rows in that XCD's L2), and a lower index has STARTED earlier.  A share is up to three pieces: the tail of a
  // tile begun by lower indices, whole tiles, and the head of a tile that higher indices complete.  The head goes FIRST
  // (its partial accumulators are published within the first few microseconds), the tail LAST (the workgroup that
  // owns a tile's last K-step adds the partials of the lower indices and runs the epilogue) -- so a wait is always for
  // a lower index, which is running (no deadlock whatever else occupies the chip), and for data that was published a
  // whole share earlier.
  struct Cur { int ph, t, k, kend; };  // ph: 0 head piece, 1 whole tiles, 2 tail piece, 3 done
  int w = blockIdx.x;
  long s_begin = 0, S_tot = 0;
  bool has_h = false, has_t = false;
  int hT = 0, hK0 = 0, hK1 = 0, f0 = blockIdx.x, f1 = total - 1, tT = 0, tK0 = 0;
  if (SK) {
    int* s_w = reinterpret_cast<int*>(smem);  // (no second LDS object beside the ring: the ring is not in use yet)
    if (tid == 0) {
      const int x = blockIdx.x & 7;
      *s_w = x * (G >> 3) + atomicAdd(&grp.sk_ctr[x], 1);
    }
    __syncthreads();
    w = __builtin_amdgcn_readfirstlane(*s_w);
    __syncthreads();
    for (int p = 0; p < grp.n; ++p)
      S_tot += (long)(grp.tile0[p + 1] - grp.tile0[p]) * ((grp.g[p].K + BK - 1) / BK);
    s_begin = (long)w * S_tot / G;
    const long s_end = (long)(w + 1) * S_tot / G;
    f0 = 0;
    f1 = -1;
    if (s_end > s_begin) {
      auto where = [&](long s, int& t, int& k, int& nt) {
        for (int p = 0; p < grp.n; ++p) {
          const long ntp = (grp.g[p].K + BK - 1) / BK, span = (long)(grp.tile0[p + 1] - grp.tile0[p]) * ntp;
          if (s < span || p == grp.n - 1) { t = grp.tile0[p] + (int)(s / ntp); k = (int)(s % ntp); nt = (int)ntp; return; }
          s -= span;
        }
      };
      int tA, kA, ntA, tB, kB, ntB;
      where(s_begin, tA, kA, ntA);
      where(s_end - 1, tB, kB, ntB);
      const bool tail = kA > 0, head = kB + 1 < ntB;
      if (tA == tB) {
        if (head) { has_h = true; hT = tA; hK0 = kA; hK1 = kB + 1; }       // head or middle piece: published
        else if (tail) { has_t = true; tT = tA; tK0 = kA; }                // tail only
        else { f0 = tA; f1 = tA; }                                         // exactly one whole tile
      } else {
        if (head) { has_h = true; hT = tB; hK0 = 0; hK1 = kB + 1; }
        if (tail) { has_t = true; tT = tA; tK0 = kA; }
        f0 = tail ? tA + 1 : tA;
        f1 = head ? tB - 1 : tB;
      }
    }
  }
  const int stride = SK ? 1 : G;
  auto enter_phase = [&](Cur& c, int ph) {  // the first phase >= ph that has work (kend of phases 1, 2: set on tile entry)
    if (SK && ph <= 0 && has_h) { c.ph = 0; c.t = hT; c.k = hK0; c.kend = hK1; return; }
    if (ph <= 1 && f0 <= f1) { c.ph = 1; c.t = f0; c.k = 0; c.kend = 0; return; }
    if (SK && ph <= 2 && has_t) { c.ph = 2; c.t = tT; c.k = tK0; c.kend = 0; return; }
    c.ph = 3;
  };
  auto step_cur = [&](Cur& c) -> bool {  // one K-step on; true when the cursor left its tile (or piece)
    if (++c.k < c.kend) return false;
    if (c.ph == 1) {
      c.t += stride;
      c.k = 0;
      if (c.t > f1) enter_phase(c, 2);
    } else {
      enter_phase(c, c.ph + 1);
    }
    return true;
  };

  struct Tile { int pi, m0, n0, nt; };
  auto locate = [&](int t, Tile& T) {
    int pi = 0;
    while (pi + 1 < grp.n && t >= grp.tile0[pi + 1]) ++pi;
    const int nwg = grp.tile0[pi + 1] - grp.tile0[pi];
    int id = t - grp.tile0[pi];
    if (!SK) {  // XCD-aware order inside a problem: ids that share an XCD (id % 8) own a contiguous run of tiles
      const int x = id & 7, q = nwg >> 3, r = nwg & 7;
      id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    }
    const int tn = grp.g[pi].tiles_n;
    const int tile_m = id / tn;
    T.pi = pi;
    T.m0 = tile_m * BM;
    T.n0 = (id - tile_m * tn) * BN;
    T.nt = (grp.g[pi].K + BK - 1) / BK;
  };

  // ---- issue cursor: the (tile, K-step) whose DMA goes out next ------------------------------------------
  Cur ci;
  enter_phase(ci, 0);
  int issued = 0;
  Tile TI{0, 0, 0, 0};
  DA da;
  DB db;
  int K_i = 0;  // (fields of the kernel-argument table are fetched by scalar loads, which share lgkmcnt with the LDS
                // reads: everything the per-step code needs is copied into registers when a cursor enters a tile)
  auto init_issue = [&]() {
    locate(ci.t, TI);
    if (ci.ph != 0) ci.kend = TI.nt;
    const GemmArgs& g = grp.g[TI.pi];
    K_i = g.K;
    da.init(g.A, reinterpret_cast<const bf16_t*>(g.A.p), TI.m0, g.M, wave, lane);
    db.init(g.B, reinterpret_cast<const bf16_t*>(g.B.p), TI.n0, g.N, wave, lane);
  };
  auto advance_issue = [&]() {
    if (ci.ph == 3) return;
    unsigned char* img = smem + (issued % NS) * STAGE;
    da.issue(img, ci.k * BK, K_i, wave);
    db.issue(img + A_BYTES, ci.k * BK, K_i, wave);
    ++issued;
    if (step_cur(ci) && ci.ph != 3) init_issue();
  };
  if (ci.ph != 3) init_issue();
  // (NS - 2 stages here, the next one at the first read_step: a refill then always targets the slot of step
  // nread - 2, whose fragment reads have been consumed by MFMAs that every wave has issued before the barrier.
  // 256-row tiles keep ONE fragment register set (two would spill): the fragments of step n are read after the MFMAs
  // of step n - 1, the slot of step n - 1 is free at the barrier of step n, so the ring runs one stage deeper)
  constexpr bool DBUF = BM <= 128;
#pragma unroll
  for (int s = 0; s < (DBUF ? NS - 2 : NS - 1); ++s) advance_issue();

  // ---- read cursor (the K-step whose fragments go LDS -> registers next) and multiply cursor ------------------
  // The fragments of step c + 1 are read while the MFMAs of step c run (two register sets, the loop body is
  // instantiated for both roles): with one set, every wave's LDS reads and MFMAs of a K-step are separate phases
  // that the per-step barrier lines up across the workgroup -- LDS phase, then MFMA phase, ~3x the MFMA time.
  struct Frag { bf16x8 a[2][TM], b[2][TN]; };
  // (rows-contiguous operands: transposed reads issued by hand -- s2st_asm.h -- and completed by finish_frags())
  struct FragRaw { typename LA::Raw a[2][TM]; typename LB::Raw b[2][TN]; };
  auto load_frags = [&](FragRaw& R, const unsigned char* cur) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < TN; ++j) R.b[s][j] = LB::raw(cur + A_BYTES, wn * WN + j * 16, s, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i) R.a[s][i] = LA::raw(cur, wm * WM + i * 16, s, lane);
    }
  };
  auto finish_frags = [&](Frag& F, FragRaw& R) {
    if (!AKM || !BKM) lds_raw_wait();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < TN; ++j) F.b[s][j] = LB::done(R.b[s][j]);
#pragma unroll
      for (int i = 0; i < TM; ++i) F.a[s][i] = LA::done(R.a[s][i]);
    }
  };
  Cur cr;
  enter_phase(cr, 0);
  int nread = 0;  // index of the step the read cursor points at
  int nt_r = 0, K_r = 0;
  auto init_read = [&]() {
    Tile T;
    locate(cr.t, T);
    if (cr.ph != 0) cr.kend = T.nt;
    nt_r = T.nt;
    K_r = grp.g[T.pi].K;
  };
  // waits for the stage of step `nread`, makes it visible to the workgroup, then reads its fragments into F
  auto read_step = [&](FragRaw& F) {
    const int ahead = issued - nread - 1;  // ring stages younger than the one about to be read
    if (ahead >= 3) S2ST_VMCNT(3 * PER_STAGE);
    else if (ahead == 2) S2ST_VMCNT(2 * PER_STAGE);
    else if (ahead == 1) S2ST_VMCNT(PER_STAGE);
    else S2ST_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    advance_issue();  // into the slot of step nread - 2: everyone's MFMAs of it have been issued (barrier above)
    unsigned char* cur = smem + (nread % NS) * STAGE;
    if (cr.k == nt_r - 1 && K_r - cr.k * BK < BK) {  // K tail: zero the invalid k of this stage
      const int kv = K_r - cr.k * BK;
      DA::sanitize(cur, kv, tid);
      DB::sanitize(cur + A_BYTES, kv, tid);
      __syncthreads();
    }
    load_frags(F, cur);
    if (!AKM || !BKM) {
      // hand-issued reads must not stay in flight across code the compiler is free to re-arrange (a tile's epilogue
      // sits between this step and its MFMAs: a register move of a not-yet-written destination would copy garbage)
      lds_raw_wait();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < TN; ++j) if (!BKM) { lds_raw_fence(F.b[s][j].lo); lds_raw_fence(F.b[s][j].hi); }
#pragma unroll
        for (int i = 0; i < TM; ++i) if (!AKM) { lds_raw_fence(F.a[s][i].lo); lds_raw_fence(F.a[s][i].hi); }
      }
    }
    ++nread;
    if (step_cur(cr) && cr.ph != 3) init_read();
  };

  Cur cc;
  enter_phase(cc, 0);
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NTHR = 64 * NW;
  // partial accumulators of a workgroup: [TM*TN][NTHR] float4, i.e. 16-byte coalesced per lane.  Hand-off between
  // workgroups (any XCDs) as cdna_hip_programming.md Guideline 16 prescribes: every storing wave drains its stores, the
  // workgroup meets, ONE lane releases at agent scope (L2 write-back), waits, then sets the flag with an agent-scope
  // atomic; the consumer polls that word relaxed with one lane, acquires at agent scope (L1 invalidate), waits, the
  // workgroup meets, and everyone reads with plain 16-byte loads.
  auto store_partial = [&]() {
    float4* dst = reinterpret_cast<float4*>(grp.sk_part + (long)w * (BM * BN));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        dst[(i * TN + j) * NTHR + tid] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    S2ST_VMCNT(0);
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      S2ST_VMCNT(0);
      __hip_atomic_store(&grp.sk_flag[w], grp.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto add_partials = [&]() {
    // contributors: w - 1, w - 2, ... while their share reaches back before this tile's first step
    const long tile_first_step = s_begin - tK0;
    for (int c = w - 1; c >= 0; --c) {
      if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(&grp.sk_flag[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != grp.epoch) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1 << 24)) break;  // bounded: a lost hand-off gives a wrong tile (caught by tests), not a hang
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        S2ST_VMCNT(0);
      }
      __syncthreads();
      const float4* src = reinterpret_cast<const float4*>(grp.sk_part + (long)c * (BM * BN));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float4 v = src[(i * TN + j) * NTHR + tid];
          acc[i][j][0] += v.x; acc[i][j][1] += v.y; acc[i][j][2] += v.z; acc[i][j][3] += v.w;
        }
      if ((long)c * S_tot / G <= tile_first_step) break;  // c began at (or before) the tile's first step
    }
  };
  Tile TC{0, 0, 0, 0};
  auto init_comp = [&]() {
    locate(cc.t, TC);
    if (cc.ph != 0) cc.kend = TC.nt;
  };
  if (cc.ph != 3) init_comp();

  Frag F;            // the operands of the K-step being multiplied
  FragRaw R0, R1_[1];  // reads in flight (the second set is dead, hence no registers, in the single-set form)
  if (cr.ph != 3) {
    init_read();
    if (DBUF) read_step(R0);
  }
  auto body = [&](FragRaw& Rc, FragRaw& Rn) {
    if (DBUF) {
      finish_frags(F, Rc);                // issued one K-step ago: landed under the previous MFMAs
      if (cr.ph != 3) read_step(Rn);      // in flight under the MFMAs below
    } else {
      read_step(Rc);
      finish_frags(F, Rc);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.b[s][j], F.a[s][i], acc[i][j], 0, 0, 0);
    const int ph = cc.ph;
    if (step_cur(cc)) {  // the last K-step of a whole tile or of a piece
      if (SK && ph == 0) {
        store_partial();  // a higher-indexed workgroup finishes this tile
      } else {
        if (SK && ph == 2) add_partials();
        gemm_epilogue<BM, BN, WGN>(grp.g[TC.pi], acc, TC.m0, TC.n0, wm, wn, lane, 0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (cc.ph != 3) init_comp();
    }
  };
  if constexpr (DBUF) {
    while (cc.ph != 3) {
      body(R0, R1_[0]);
      if (cc.ph == 3) break;
      body(R1_[0], R0);
    }
  } else {
    while (cc.ph != 3) body(R0, R0);
  }
  if (SK) {  // the last workgroup to leave re-arms the counters for the next launch on this stream
    __syncthreads();
    if (tid == 0 && atomicAdd(&grp.sk_ctr[8], 1) == G - 1) {
#pragma unroll
      for (int x = 0; x < 9; ++x) grp.sk_ctr[x] = 0;
    }
  }
}

template <int BM, int BN, bool AKM, bool BKM, int NS, int NW>
const char* persistent_tag() {
  static char buf[96];
  if (!buf[0])
    snprintf(buf, sizeof buf, "gemm_bf16_dma_persistent_kernel<%d, %d, %s, %s, %d, %d, false>", BM, BN, AKM ? "true" : "false",
             BKM ? "true" : "false", NS, NW);
  return buf;
}

template <int BM, int BN, bool AKM, bool BKM, int NS, int NW>
const char* streamk_tag() {
  static char buf[104];
  if (!buf[0])
    snprintf(buf, sizeof buf, "gemm_bf16_dma_persistent_kernel<%d, %d, %s, %s, %d, %d, true>", BM, BN, AKM ? "true" : "false",
             BKM ? "true" : "false", NS, NW);
  return buf;
}

#endif  // S2ST_EXPERIMENTAL

// S2ST_GEMM_DMA=0: the register-staged kernel everywhere (no LDS-DMA ring) -- the A/B switch of round 1, read once
int gemm_dma_enabled() {
  static const int v = s2st_env_int("S2ST_GEMM_DMA", 1);
  return v;
}

int num_cus() {
  // per device (a process drives one GPU, but which one is hipSetDevice's business, not device 0's: VERDICT r5 item 7c)
  static int n[16] = {0};
  static const int forced = s2st_env_int("S2ST_GEMM_PERSIST_WGS", 0) > 0 ? s2st_env_int("S2ST_GEMM_PERSIST_WGS", 0) : 0;  // tuning aid
  if (forced) return forced;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (!n[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n[dev] = v;
  }
  return n[dev];
}

// ---- stream-K scratch per stream (bound by the training engine for its two streams at the start of a step) ----------
struct SkScratch { hipStream_t st; float* p; long floats; };
SkScratch g_sk[4];
int g_sk_n = 0;
int g_sk_epoch = 0;

// S2ST_GEMM_STREAMK: 0 = whole tiles per workgroup, 1 = stream-K where a scratch buffer is bound to the stream.  On by
// default for callers that bind a scratch themselves (s2st_gemm_streamk_scratch); the training engine binds one only
// with S2ST_GEMM_STREAMK=1 in the environment: measured on MI355X (tools/streamk_probe.py) the
// hand-off -- L2 write-back behind the release, invalidate + 64 KB read behind the acquire -- costs 12-20 us per launch,
// more than the idle last round of the 25-45 us products of a training step; it pays from ~2 rounds of long-K tiles on.
#ifdef S2ST_EXPERIMENTAL
int streamk_mode() {
  const char* ev = s2st_env_str("S2ST_GEMM_STREAMK");
  return ev ? atoi(ev) : 1;
}

template <int BM, int BN, int NS, int NW>
int launch_persistent(const GemmGroup& grp_in, hipStream_t st) {
  constexpr int LDS = NS * (BM + BN) * 128;
  GemmGroup grp = grp_in;
  // stream-K when a scratch buffer is bound to this stream and the split pays: uneven rounds of whole tiles (or fewer
  // tiles than CUs) and enough K-steps per workgroup that the partial-sum hand-off (64 KB out, 64 KB in) is small change
  grp.sk = 0;
  if (streamk_mode() > 0 && BM * BN == 128 * 128 && grp.total > 0) {
    SkScratch* sc = nullptr;
    for (int i = 0; i < g_sk_n; ++i) if (g_sk[i].st == st && g_sk[i].p) sc = &g_sk[i];
    long steps = 0;
    for (int i = 0; i < grp.n; ++i) steps += (long)(grp.tile0[i + 1] - grp.tile0[i]) * ((grp.g[i].K + BK - 1) / BK);
    const int G = num_cus() & ~7;
    const long rounds = (grp.total + G - 1) / G;
    const double waste = 1.0 - (double)grp.total / (double)(rounds * G);  // idle share of the last round
    const int min_steps = s2st_env_int("S2ST_STREAMK_MIN_STEPS", 8);
    if (sc && G >= 8 && G <= S2ST_STREAMK_MAX_WGS && waste > 0.08 && steps >= (long)min_steps * G &&
        sc->floats >= S2ST_STREAMK_SCRATCH_FLOATS) {
      grp.sk = 1;
      grp.epoch = ++g_sk_epoch;
      if (g_sk_epoch > (1 << 30)) g_sk_epoch = 0;
      grp.sk_ctr = reinterpret_cast<int*>(sc->p);
      grp.sk_flag = reinterpret_cast<int*>(sc->p) + 16;
      grp.sk_part = sc->p + 1024;
    }
  }
  double fl = 0, by = 0;
  for (int i = 0; i < grp.n; ++i) { fl += gemm_flops(grp.g[i]); by += gemm_min_bytes(grp.g[i]); }
  // S2ST_GROUP_WGS=<n> (tuning aid): cap for grouped launches, which share the chip with the data-path stream
  constexpr int group_cap = 0;  // (round 2's S2ST_GROUP_WGS: capping the weight-gradient launch to fewer CUs made it the critical path)
  const int cap = (grp.n > 1 && group_cap > 0) ? group_cap : num_cus();
  const int grid = grp.sk ? (num_cus() & ~7) : (grp.total < 1 ? 1 : (grp.total < cap ? grp.total : cap));  // (preload: no tiles)
  auto go = [&](auto kern, const char* tag) {
    static bool configured = false;
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
        return -1;
      configured = true;
    }
    s2st_launch(tag, fl, by, kern, dim3(grid), dim3(64 * NW), LDS, st, grp);
    return 0;
  };
  const bool akm = grp.g[0].A.kmajor != 0, bkm = grp.g[0].B.kmajor != 0;
  if (grp.sk) {
    if constexpr (BM == 128 && BN == 128) {
      if (akm && bkm) return go(gemm_bf16_dma_persistent_kernel<BM, BN, true, true, NS, NW, true>, streamk_tag<BM, BN, true, true, NS, NW>());
      if (akm && !bkm) return go(gemm_bf16_dma_persistent_kernel<BM, BN, true, false, NS, NW, true>, streamk_tag<BM, BN, true, false, NS, NW>());
      if (!akm && bkm) return go(gemm_bf16_dma_persistent_kernel<BM, BN, false, true, NS, NW, true>, streamk_tag<BM, BN, false, true, NS, NW>());
      return go(gemm_bf16_dma_persistent_kernel<BM, BN, false, false, NS, NW, true>, streamk_tag<BM, BN, false, false, NS, NW>());
    }
  }
  if (akm && bkm) return go(gemm_bf16_dma_persistent_kernel<BM, BN, true, true, NS, NW>, persistent_tag<BM, BN, true, true, NS, NW>());
  if (akm && !bkm) return go(gemm_bf16_dma_persistent_kernel<BM, BN, true, false, NS, NW>, persistent_tag<BM, BN, true, false, NS, NW>());
  if (!akm && bkm) return go(gemm_bf16_dma_persistent_kernel<BM, BN, false, true, NS, NW>, persistent_tag<BM, BN, false, true, NS, NW>());
  return go(gemm_bf16_dma_persistent_kernel<BM, BN, false, false, NS, NW>, persistent_tag<BM, BN, false, false, NS, NW>());
}

#else
int streamk_mode() { return 0; }
template <int BM, int BN, int NS, int NW>
int launch_persistent(const GemmGroup&, hipStream_t) { return S2ST_ERR_ARG; }  // (not built: never reached, see persist_mode())
#endif  // S2ST_EXPERIMENTAL

// C(m, n) (+)= sum_s slab[z][s][m][n]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, GemmOut C, int M, int N,
                                                            int splitk, int zdiv, int accumulate) {
  const int nq = (N + 3) >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * nq) return;
  const int m = (int)(i / nq), n = (int)(i - (long)m * nq) * 4;
  const int z = blockIdx.y, zq = z / zdiv, zr = z - zq * zdiv;
  const float* sp = slab + ((long)z * splitk * M + m) * N + n;
  float* cp = C.p + zq * C.zo + zr * C.zi + split_off(C.sp, m) + n;
  const long sstride = (long)M * N;
  if ((N & 3) == 0 && ((uintptr_t)cp & 15) == 0) {
    // slab loads are issued 8 at a time (a load-add chain costs one memory latency per slab); adds in slab order
    float4 a = accumulate ? *reinterpret_cast<const float4*>(cp) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sb = 0; sb < splitk; sb += 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(sp + (long)min(sb + j, splitk - 1) * sstride);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (sb + j < splitk) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
    }
    *reinterpret_cast<float4*>(cp) = a;
  } else {
    for (int r = 0; r < 4 && n + r < N; ++r) {
      float a = accumulate ? cp[r] : 0.f;
      for (int s = 0; s < splitk; ++s) a += sp[s * sstride + r];
      cp[r] = a;
    }
  }
}

template <int BM, int BN, bool VEC>
void launch_layouts(const GemmArgs& g, dim3 grid, hipStream_t st) {
  const double fl = gemm_flops(g), by = gemm_min_bytes(g);
  if (g.A.kmajor && g.B.kmajor)
    s2st_launch(staged_tag<BM, BN, true, true, VEC>(), fl, by, gemm_bf16_kernel<BM, BN, true, true, VEC>, grid, dim3(256), 0, st, g);
  else if (g.A.kmajor && !g.B.kmajor)
    s2st_launch(staged_tag<BM, BN, true, false, VEC>(), fl, by, gemm_bf16_kernel<BM, BN, true, false, VEC>, grid, dim3(256), 0, st, g);
  else if (!g.A.kmajor && g.B.kmajor)
    s2st_launch(staged_tag<BM, BN, false, true, VEC>(), fl, by, gemm_bf16_kernel<BM, BN, false, true, VEC>, grid, dim3(256), 0, st, g);
  else
    s2st_launch(staged_tag<BM, BN, false, false, VEC>(), fl, by, gemm_bf16_kernel<BM, BN, false, false, VEC>, grid, dim3(256), 0, st, g);
}

// 16-byte fast path: aligned base, ld and batch strides multiples of 8 elements.  (K tails are
// masked in the kernel; rows-contiguous operands rely on the padded-row contract.)
bool vec_ok(const GemmOperand& o) {
  return ((uintptr_t)o.p % 16 == 0) && (o.sp.ld % 8 == 0) && (o.sp.bs % 8 == 0) && (o.zo % 8 == 0) &&
         (o.zi % 8 == 0);
}

bool dma_layout_ok(const GemmArgs& g) { return (g.A.kmajor || g.A.sp.per <= 0) && (g.B.kmajor || g.B.sp.per <= 0); }

// alignment flags the kernels / epilogue read; returns whether the 16-byte operand path applies
bool prep_flags(GemmArgs& g) {
  const bool vec = vec_ok(g.A) && vec_ok(g.B);
  g.avec = g.bvec = vec ? 1 : 0;
  g.cvec = ((!g.C.p || (uintptr_t)g.C.p % 16 == 0) && (!g.C.h || (uintptr_t)g.C.h % 8 == 0) &&
            g.C.sp.ld % 4 == 0 && g.C.sp.bs % 4 == 0 && g.C.zo % 4 == 0 && g.C.zi % 4 == 0 &&
            (!g.ep.resid || (uintptr_t)g.ep.resid % 16 == 0) && (!g.ep.bias || ((uintptr_t)g.ep.bias % 16 == 0 && g.ep.bias_zo % 4 == 0)))
               ? 1 : 0;
  return vec;
}

// bit 1 of cvec: the straight-line epilogue applies (gemm_epilogue_fast; S2ST_GEMM_FAST_EPI=0 is the A/B switch).
// Called once split-K / slab decisions are made.
void mark_fast_epilogue(GemmArgs& g) {
  g.cvec &= 1;  // (bits 1 .. 3 are set below)
  const char* ev = s2st_env_str("S2ST_GEMM_FAST_EPI");
  if (ev && atoi(ev) == 0) return;
  const bool drop = g.ep.drop_p > 0.f, res = g.ep.resid != nullptr, accu = g.ep.accumulate != 0;
  const bool combo = !accu || (!drop && !res && !g.C.h);  // instantiated: {-, resid, drop, drop + resid} x outputs, accumulate (fp32)
  // the bf16 copy is stored 16 bytes (8 columns) per lane
  const bool h16 = !g.C.h || ((uintptr_t)g.C.h % 16 == 0 && g.C.sp.ld % 8 == 0 && g.C.sp.bs % 8 == 0 && g.C.zo % 8 == 0 &&
                              g.C.zi % 8 == 0 && g.N % 8 == 0);
  if (g.cvec && g.N >= 4 && g.N % 4 == 0 && g.M >= 1 && g.splitk == 1 && !g.slab && !g.ep.mask_y && !g.ep.colsum && !g.ep.colsum_part &&
      (g.ep.act != 2 || (!drop && !accu)) && combo && h16 && (g.C.p || g.C.h) && (!accu || g.C.p))
    g.cvec |= 2;
  // bit 3: the masked data-gradient form (bf16 output only, nothing else in the epilogue but the column sums)
  if (g.cvec && g.ep.mask_y && g.C.h && !g.C.p && h16 && g.N >= 8 && g.M >= 1 && g.splitk == 1 && !g.slab && !g.ep.bias &&
      g.ep.act == 0 && !drop && !res && !accu && (uintptr_t)g.ep.mask_y % 16 == 0)
    g.cvec |= 8;
}

// S2ST_GEMM_PERSIST: 0 = one-shot kernels only, 1 = grouped launches persistent, single products one-shot (default:
// with more tiles than CUs the hardware dispatcher back-fills CUs as one-shot workgroups retire, which balances better
// than a fixed walk of 2.25 tiles per workgroup -- 10.05 vs 10.58 ms per training step, profiles/r02_ab_switches.txt),
// 2 = every 128-row launch persistent, 3 = grouped launches and single products with more tiles than CUs
// CUs a single product's tiles are priced over (the tile pickers' round counts)
int data_cus() { return num_cus(); }

int persist_mode() {
  const char* ev = s2st_env_str("S2ST_GEMM_PERSIST");  // read per call: an A/B switch the tests flip
  const int v = ev ? atoi(ev) : 1;
#ifdef S2ST_EXPERIMENTAL
  return v;
#else
  return v > 1 ? 1 : v;  // (2 / 3 need the persistent kernels: experimental builds only)
#endif
}

constexpr bool kExperimental =
#ifdef S2ST_EXPERIMENTAL
    true;
#else
    false;
#endif

// S2ST_GEMM_W4 (read per call: an A/B switch): 0 = the 8-wave ring kernels only (one workgroup per CU); -1 (default) =
// the 4-wave early-release form (gemm_bf16_w4.hip, 2 - 3 workgroups per CU) for single products with at least one
// 128 x 128 tile per CU, see w4_pick(); 1 = every 128-row single product on it; 2 = grouped weight gradients too
int w4_mode() {
  const char* ev = s2st_env_str("S2ST_GEMM_W4");
  return ev ? atoi(ev) : -1;
}

// Which form runs a single product (measured per shape on MI355X with tools/gemm_forms_bench.py, kernel time from
// events attached to the dispatch, profiles/r03_gemm_forms.txt).  With fewer 128 x 128 tiles than CUs (M ~ 4.6 k tokens x
// N = 512: 144 tiles) nothing co-resides anyway and the 8-wave kernel's faster K-loop wins (17.5 vs 23.9 us at K = 2048);
// from one tile per CU on, two or three co-resident workgroups hide each other's prologue / epilogue / barrier waits:
// N = 1536 ... 2048 at K = 512 run 10 - 30 % shorter (4584 x 2048 x 512: 17.5 vs 21.4 us; 2800 x 1536 x 512: 11.1 vs 15.8).
// Between the two 4-wave shapes the fuller last round wins: 128 x 128 tiles over 2 slots per CU against 128 x 64 tiles
// over 3 (the pick agreed with the faster of the two on every measured shape).
// Returns 0 (8-wave kernels), else bn of the 128-row 4-wave form.
int w4_pick(const GemmArgs& g, bool dma_ok) {
  const int mode = w4_mode();
  if (mode == 0 || !dma_ok || g.M < 128) return 0;
  const long tm = (g.M + 127) / 128;
  const long t128 = tm * ((g.N + 127) / 128) * g.batch, t64 = tm * ((g.N + 63) / 64) * g.batch;
  if (mode < 0 && t128 < data_cus()) return 0;
  const long s128 = 2L * data_cus(), s64 = 3L * data_cus();
  const double e128 = (double)t128 / (double)(((t128 + s128 - 1) / s128) * s128);
  const double e64 = (double)t64 / (double)(((t64 + s64 - 1) / s64) * s64);
  // Round 5: many rounds of long K-loops are priced by the steady state, not by the last round's fill -- per flop the
  // 128 x 128 tile needs 2/3 of the LDS reads of the 128 x 64 one.  HuBERT's conv stack (307 k / 154 k rows x 512 x 1536):
  // 632 vs 750 us and 342 vs 402 us, where the fill rule took 128 x 64 (25.0 full rounds against 18.75);
  // tools/conv_forms_bench.py, profiles/r05_hubert_forms.txt.  (The training step's products have < 4 rounds: unchanged.)
  if (g.N > 64 && g.K >= 1024 && t128 >= 4 * s128) return 128;
  return (g.N > 64 && e128 >= e64) ? 128 : 64;
}

// S2ST_GEMM_P4 (read per call: an A/B switch): 0 = never; -1 (default) = the 256 x 256 four-phase form (gemm_bf16_p4.hip)
// where p4_pick() says so; 1 = every product it can run.  It can run: both operands K-contiguous with plain row strides,
// 16-byte aligned, batch 1, no split-K, no masked epilogue.  The pick is a two-line cost model fitted to
// tools/gemm_p4_bench.py (profiles/r05_gemm_p4_bench.txt, random operands, kernel time from events on the dispatch):
//   four-phase form : rounds of 256-row tiles over the CUs x (2.0 us per 64-deep K-tile + 7 us of prologue / epilogue)
//   128-row forms   : ~650 TFLOP/s on products of this size (577 - 693 measured; 870 on 4096^3)
// It takes 4096^3 (126 vs 159 us: 1.09 PFLOP/s), HuBERT's FFN-in products (9600 x 3072 x 768: 64 vs 78 us; 19200 rows: 119 vs
// 148) and leaves the K = 512 products of the training step (one round of 8 K-tiles: 21 vs 14 - 19 us) and the short-N
// ones (768 columns: a third of the CUs) where they were.
int p4_mode() {
  const char* ev = s2st_env_str("S2ST_GEMM_P4");
  return ev ? atoi(ev) : -1;
}
bool p4_can(const GemmArgs& g, bool dma_ok) {
  return dma_ok && g.A.kmajor && g.B.kmajor && g.batch == 1 && g.A.sp.per <= 0 && g.B.sp.per <= 0 && !g.ep.mask_y &&
         !g.ep.colsum && !g.ep.colsum_part;
}
bool p4_pick(const GemmArgs& g, bool dma_ok) {
  const int mode = p4_mode();
  if (mode == 0 || !p4_can(g, dma_ok)) return false;
  if (mode >= 1) return true;
  if (g.M < 512 || g.N < 512 || g.K < 8 * BK) return false;
  const long tiles = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
  const long ncu = data_cus();
  const long rounds = (tiles + ncu - 1) / ncu;
  const double t_p4 = (double)rounds * (2.0 * ((g.K + BK - 1) / BK) + 7.0);          // us
  const double t_ring = 2.0 * g.M * (double)g.N * g.K / 650e6;                        // us at 650 TFLOP/s
  // (0.80, not parity: beside a training step -- HuBERT's front end runs ahead on a second stream -- a 512-thread workgroup
  // holding 128 KB of LDS for tens of microseconds keeps the step's short dependent kernels off its CU; with the threshold
  // at 0.92 the front end's FFN-in products took this form, 64 vs 78 us alone, and config 3 / 4 got SLOWER: 10.77 - 11.0
  // vs 10.63 ms per step, profiles/r05_hubert_p4_ab.txt)
  return t_p4 < 0.80 * t_ring;
}

template <int BN, int BM = 128>
void add_to_group(GemmGroup& grp, GemmArgs g) {
  g.splitk = 1;
  g.slab = nullptr;
  g.kchunk = ((g.K + BK - 1) / BK) * BK;
  g.tiles_n = (g.N + BN - 1) / BN;
  mark_fast_epilogue(g);
  const int i = grp.n++;
  grp.g[i] = g;
  grp.tile0[i + 1] = grp.tile0[i] + ((g.M + BM - 1) / BM) * g.tiles_n;
  grp.total = grp.tile0[i + 1];
}

}  // namespace

void s2st_gemm_streamk_bind(hipStream_t st, float* scratch, long floats) {
  for (int i = 0; i < g_sk_n; ++i)
    if (g_sk[i].st == st) { g_sk[i].p = scratch; g_sk[i].floats = floats; return; }
  if (g_sk_n < 4) g_sk[g_sk_n++] = SkScratch{st, scratch, floats};
}
void s2st_gemm_streamk_unbind_all() { g_sk_n = 0; }

bool s2st_gemm_group_ok(const GemmArgs& g0) {
  GemmArgs g = g0;
  if (g.A.dtype != S2ST_BF16 || g.B.dtype != S2ST_BF16 || g.precise || persist_mode() == 0) return false;
  static const int use_dma = gemm_dma_enabled();
  if (!use_dma || !prep_flags(g) || !dma_layout_ok(g) || g.batch != 1 || g.M < 128 || g.N < 128 || g.K < 1) return false;
  if (g.ep.mask_y && (!g.cvec || g.N % 4 != 0)) return false;
  return true;
}

// list[0..n): problems that passed s2st_gemm_group_ok, all with the layouts of list[0]
int s2st_gemm_bf16_group(const GemmArgs* list, int n, hipStream_t st) {
  if (n <= 0) return 0;
  if (n > S2ST_GROUP_MAX) return S2ST_ERR_ARG;
  GemmGroup grp{};
  for (int i = 0; i < n; ++i) {
    GemmArgs g = list[i];
    if (g.zdiv <= 0) g.zdiv = 1;
    if (!prep_flags(g) || g.A.kmajor != list[0].A.kmajor || g.B.kmajor != list[0].B.kmajor) return S2ST_ERR_ARG;
    add_to_group<128>(grp, g);
  }
  // S2ST_GROUP_TILE=256: 256 x 128 tiles (48 KB of operands per K-step for twice the FLOPs: half the workgroups, which
  // leaves CUs to the data-path stream the group runs beside)
  const char* gt = s2st_env_str("S2ST_GROUP_TILE");
  bool big = kExperimental && gt && atoi(gt) == 256;
  for (int i = 0; i < n && big; ++i) big = list[i].M >= 256;
  if (big) {
    GemmGroup g2{};
    for (int i = 0; i < n; ++i) {
      GemmArgs g = list[i];
      if (g.zdiv <= 0) g.zdiv = 1;
      prep_flags(g);
      add_to_group<128, 256>(g2, g);
    }
    if constexpr (kExperimental) {
      if (launch_persistent<256, 128, 3, 8>(g2, st)) return S2ST_ERR_LAUNCH;
    }
    return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
  }
  // default: one workgroup per tile of the concatenated list (plain K-loop: 9.65 vs 9.76 ms/step);
  // S2ST_GROUP_ONESHOT=0: the persistent tile walk (also what S2ST_GROUP_TILE=256 and a bound stream-K scratch use)
  const char* os = s2st_env_str("S2ST_GROUP_ONESHOT");
  bool sk_bound = false;
  for (int i = 0; i < g_sk_n; ++i) sk_bound = sk_bound || (g_sk[i].st == st && g_sk[i].p);
  if (!kExperimental || (!(os && atoi(os) == 0) && !(sk_bound && streamk_mode() > 0))) {
    // S2ST_GROUP_XCD=0 (A/B switch): every product's tiles spread over all XCDs (the form up to round 3)
    constexpr bool xcd_global = true;  // (one contiguous run of the tile list per XCD: 206 -> 121 MB fetched per launch, round 3)
    grp.xcd_global = xcd_global ? 1 : 0;
    if (w4_mode() >= 2 ? s2st_gemm_bf16_w4_group(grp, st) : launch_dma_group<128, 128, 4, 8, true>(grp, st)) return S2ST_ERR_LAUNCH;
    return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
  }
  if (launch_persistent<128, 128, 4, 8>(grp, st)) return S2ST_ERR_LAUNCH;
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

// Launch every kernel instantiation once on an empty problem (M = N = K = 0: no loads, no stores) so
// that code objects are resident and the > 64 KiB dynamic-LDS attribute is set before the first
// timed step (first use of a kernel on a fresh process costs milliseconds).
int s2st_gemm_bf16_preload(hipStream_t st) {
  GemmArgs g{};
  g.A.dtype = g.B.dtype = S2ST_BF16;
  g.splitk = 1; g.zdiv = 1; g.tiles_n = 1; g.batch = 1; g.kchunk = BK;
  dim3 grid(1, 1, 1);
  int rc = 0;
  for (int lay = 0; lay < 4; ++lay) {
    g.A.kmajor = lay & 1; g.B.kmajor = (lay >> 1) & 1;
    launch_layouts<128, 128, true>(g, grid, st);
    launch_layouts<128, 64, true>(g, grid, st);
    launch_layouts<64, 64, true>(g, grid, st);
    launch_layouts<64, 64, false>(g, grid, st);
    rc |= launch_dma<128, 128, 4, 8, true>(g, grid, st);
    rc |= launch_dma<128, 64, 4, 8, true>(g, grid, st);
    rc |= launch_dma<64, 64, 4, 4, true>(g, grid, st);
    GemmGroup grp{};  // no tiles: the kernels fall straight through
    grp.n = 1;
    grp.g[0] = g;
    if constexpr (kExperimental) {
      rc |= launch_persistent<128, 128, 4, 8>(grp, st);
      rc |= launch_persistent<128, 64, 4, 8>(grp, st);
      rc |= launch_persistent<256, 128, 3, 8>(grp, st);
      rc |= launch_dma<256, 128, 3, 8>(g, grid, st);
    }
    { GemmGroup g0 = grp; g0.total = 0; g0.n = 1; rc |= launch_dma_group<128, 128, 4, 8, true>(g0, st); }
  }
  rc |= s2st_gemm_bf16_w4_preload(st);
  rc |= s2st_gemm_bf16_p4_preload(st);
  return rc || hipGetLastError() != hipSuccess ? -1 : 0;
}

// tile choice: estimated time = rounds over the 256 CUs x per-tile work / per-tile efficiency
int s2st_gemm_bf16(GemmArgs g, hipStream_t st, int* bm_out) {
  const bool vec = prep_flags(g);
  if (g.ep.mask_y && (!g.cvec || g.N % 4 != 0)) return S2ST_ERR_SHAPE;
  const bool linear_epi = !g.ep.act && g.ep.drop_p == 0.f && !g.ep.mask_y;
  struct Cand { int bm, bn; double eff; };
  static const Cand cands[3] = {{128, 128, 1.0}, {128, 64, 0.8}, {64, 64, 0.55}};  // (tile-preference weights: round 2's sweep)
  int bm = 64, bn = 64;
  if (vec) {
    double best = 1e300;
    for (const Cand& c : cands) {
      long tiles = (long)((g.M + c.bm - 1) / c.bm) * ((g.N + c.bn - 1) / c.bn) * g.batch;
      const long ncu = data_cus();
      long rounds = (tiles + ncu - 1) / ncu;
      // split-K candidates fill the chip anyway: cost by work / efficiency only
      double cost = (double)rounds * c.bm * c.bn / c.eff;
      if (cost < best) { best = cost; bm = c.bm; bn = c.bn; }
    }
  }
  if (vec) {  // tuning aid: S2ST_GEMM_TILE=128x128|128x64|64x64 forces the tile
    const char* force = s2st_env_str("S2ST_GEMM_TILE");  // (read per call: the tests switch it)
    if (force && sscanf(force, "%dx%d", &bm, &bn) != 2) { bm = 64; bn = 64; }
  }
  const bool forced_p4 = vec && bm == 256 && bn == 256;  // S2ST_GEMM_TILE=256x256: the four-phase form where it can run
  if (bm == 256 && !forced_p4 && !(kExperimental && vec && dma_layout_ok(g) && g.batch == 1)) { bm = 128; bn = 128; }  // 256-row tiles: ring kernels of experimental builds only
  const bool can_split_ = g.ep.accumulate && linear_epi && g.C.p && !g.C.h && !g.ep.bias && !g.ep.resid && g.K >= 8 * BK;
  if (vec && can_split_ && (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.batch < 256) { bm = 128; bn = 128; }
  // the 4-wave early-release form (w4_pick above): a forced tile is honoured (S2ST_GEMM_W4 >= 1 puts it on that form)
  int w4bn = 0;
  {
    static const int use_dma_w4 = gemm_dma_enabled();
    const bool split_like = can_split_ && (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.batch < 256;
    if (vec && use_dma_w4 && dma_layout_ok(g) && !split_like && persist_mode() != 2) {
      if (s2st_env_str("S2ST_GEMM_TILE")) w4bn = (w4_mode() >= 1 && bm == 128 && (bn == 128 || bn == 64)) ? bn : 0;
      else w4bn = w4_pick(g, true);
      if (w4bn) { bm = 128; bn = w4bn; }
    }
  }
  // the 256 x 256 four-phase form (p4_pick above); a forced tile is honoured where the form can run at all
  bool p4 = false;
  {
    static const int use_dma_p4 = gemm_dma_enabled();
    const bool split_like = can_split_ && (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.batch < 256;
    const bool can = vec && use_dma_p4 && !split_like && persist_mode() != 2 && p4_can(g, dma_layout_ok(g));
    if (s2st_env_str("S2ST_GEMM_TILE")) p4 = forced_p4 && can && p4_mode() != 0;
    else p4 = can && p4_pick(g, true);
    if (forced_p4 && !p4) { bm = 128; bn = 128; }
    if (p4) { bm = 256; bn = 256; w4bn = 0; }
  }
  // (a forced 256 x 256 form on a product that then splits K -- fewer than 256 tiles of 256 x 256 -- used to fall
  //  through to the 64 x 64 launch with a grid built for 256 x 256 tiles: most of C never written.  The four-phase kernel has
  //  no split-K form, so such a product goes back to 128 x 128 tiles BEFORE the grid is derived.)
  if (p4 && can_split_ && (long)((g.M + 255) / 256) * ((g.N + 255) / 256) * g.batch < 256) { p4 = false; bm = 128; bn = 128; }
  const int tm = (g.M + bm - 1) / bm, tn = (g.N + bn - 1) / bn;
  const long nt = (long)tm * tn * g.batch;
  // split-K only for accumulating fp32 outputs with a linear epilogue (weight gradients): K is
  // split over workgroups; partial sums go to fp32 slabs in the caller's scratch and are combined
  // by splitk_reduce_kernel (HBM/MALL streaming, ~5 TB/s), or -- without scratch -- are added
  // with fp32 atomics (~1.3 TB/s chip-wide, so only a fallback).
  g.splitk = 1;
  g.slab = nullptr;
  if (can_split_ && nt < 256) {
    static const int target = s2st_env_int("S2ST_SPLITK_TARGET", 128);
    int want = (int)((target + nt - 1) / nt);
    int maxs = g.K / (4 * BK);
    g.splitk = want < maxs ? want : maxs;
    if (g.ws) {
      long fit = g.ws_floats / ((long)g.M * g.N * g.batch);
      if (fit < g.splitk) g.splitk = (int)fit;
    }
    if (g.splitk < 1) g.splitk = 1;
  }
  const int kt = (g.K + BK - 1) / BK;
  g.kchunk = ((kt + g.splitk - 1) / g.splitk) * BK;
  g.splitk = (g.K + g.kchunk - 1) / g.kchunk;
  if (g.splitk < 1) g.splitk = 1;
  g.tiles_n = tn;
  const bool use_slab = g.splitk > 1 && g.ws != nullptr;
  if (use_slab) g.slab = g.ws;
  mark_fast_epilogue(g);
  dim3 grid(tm * tn, g.batch * g.splitk, 1);
  if (grid.y > 65535) return -2;
  if (bm_out) *bm_out = bm * 1000 + bn;
  // LDS-DMA ring kernel: aligned operands; rows-contiguous operands need a plain k stride
  static const int use_dma = gemm_dma_enabled();
  const bool dma_ok = vec && use_dma && dma_layout_ok(g);
  // more tiles than CUs: the persistent kernel keeps the DMA ring running across the tiles a workgroup walks
  // ... and, with a stream-K scratch bound to the stream, also the 128 x 128 launches that leave CUs idle (N = 512
  // layers: 144 tiles) when K is long enough to share (launch_persistent decides)
  bool sk_bound = false;
  if (streamk_mode() > 0 && bm == 128 && bn == 128)
    for (int i = 0; i < g_sk_n; ++i) sk_bound = sk_bound || (g_sk[i].st == st && g_sk[i].p);
  if (dma_ok && g.batch == 1 && g.splitk == 1 && bm == 128 && persist_mode() > 0 && !w4bn &&
      (persist_mode() == 2 || (persist_mode() == 3 && nt > num_cus()) || (sk_bound && (nt > num_cus() || (long)nt * ((g.K + BK - 1) / BK) >= 8L * num_cus())))) {
    GemmGroup grp{};
    int rc;
    if (bn == 128) { add_to_group<128>(grp, g); rc = launch_persistent<128, 128, 4, 8>(grp, st); }
    else { add_to_group<64>(grp, g); rc = launch_persistent<128, 64, 4, 8>(grp, st); }
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (dma_ok) {
    int rc;
    // (one instantiation per tile shape: the interleaved steady state, 8 waves for 128-row tiles; the round-2 A/B forms
    // -- plain loop, 4 waves, 2 / 3 / 5 ring stages, 64 x 128 -- were measured then and are no longer built)
    if (p4 && g.splitk == 1) rc = s2st_gemm_bf16_p4(g, grid, st);
    else if (w4bn && g.splitk == 1) rc = s2st_gemm_bf16_w4(g, bm, bn, grid, st);
    else if (kExperimental && bm == 256 && bn == 128) { if constexpr (kExperimental) rc = launch_dma<256, 128, 3, 8>(g, grid, st); else rc = -1; }
    else if (bm == 128 && bn == 128) rc = launch_dma<128, 128, 4, 8, true>(g, grid, st);
    else if (bm == 128) rc = launch_dma<128, 64, 4, 8, true>(g, grid, st);
    else rc = launch_dma<64, 64, 4, 4, true>(g, grid, st);
    if (rc) return rc;
  } else
  if (!vec) launch_layouts<64, 64, false>(g, grid, st);
  else if (bm == 128 && bn == 128) launch_layouts<128, 128, true>(g, grid, st);
  else if (bm == 128) launch_layouts<128, 64, true>(g, grid, st);
  else launch_layouts<64, 64, true>(g, grid, st);
  if (use_slab) {
    const long nthr = (long)g.M * ((g.N + 3) / 4);
    // bytes: every slab once, the result once (twice when it accumulates)
    const double by = 4.0 * g.M * g.N * g.batch * (g.splitk + 1 + (g.ep.accumulate ? 1 : 0));
    s2st_launch("splitk_reduce_kernel", by, 0.0, splitk_reduce_kernel, dim3((unsigned)((nthr + 255) / 256), g.batch),
                dim3(256), 0, st, (const float*)g.slab, g.C, g.M, g.N, g.splitk, g.zdiv, g.ep.accumulate);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------
// Skinny-M product for AR decoding (M = utterances of the batch, <= 16): y[M][N] = f(x[M][K] W[N][K]^T + b) (+ resid).
// The tiled kernels above spend a whole 64x64 (or larger) tile, a DMA ring and a cast pass of x on 16 rows; here a
// workgroup owns 16 output columns, its 4 waves split K (interleaved 32-wide steps), each wave reads its slice of
// the 16 weight rows straight from HBM/L2 (16 bytes per lane, all loads of 4 steps in flight) and converts the
// fp32 activations to bf16 in registers -- the same rounding the cast kernel applies, so every product is the one
// the tiled path forms; only the fp32 summation order over K differs.  Partial accumulators meet in LDS.
// MFMA roles: a = weight rows (-> accumulator rows n), b = activation rows (-> accumulator columns m), i.e. lane l
// holds y[m = l & 15][n0 + 4 (l >> 4) + r], the layout of gemm_epilogue above (float4 stores along n).
// MT = row blocks of 16 (M <= 16 MT): round 4 decodes up to 64 utterances per step (the reference's generate_waveform
// batches by --max-tokens 100000, run_baseline.sh:147), so a lane keeps MT accumulators and the weight fragment it
// fetched is used MT times.  act: 0 none, 1 ReLU, 2 GELU, 3 logistic (the stop head: sigmoid fused, speech_generator_for_s2st.py:91).
template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const float* __restrict__ A, long lda,
                                                          const bf16_t* __restrict__ W, long ldw,
                                                          float* __restrict__ C, long ldc,
                                                          const float* __restrict__ bias, int act, float drop_p,
                                                          uint64_t seed, const float* __restrict__ resid, long ldr,
                                                          int M, int N, int K, const float* __restrict__ ln_g,
                                                          const float* __restrict__ ln_b, float ln_eps,
                                                          const uint64_t* __restrict__ seed_ptr,
                                                          const int* __restrict__ row_map) {
  if (seed_ptr) seed = *seed_ptr;  // (replayable decode step: the step's seeds live in device memory)
  __shared__ __attribute__((aligned(16))) float part[3][MT * 64 * 4];
  __shared__ float ln_mean[16 * MT], ln_rstd[16 * MT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n0 = blockIdx.x * 16;
  const int kc = (lane >> 4) * 8;
  {
    // blockIdx.y: the block of 16 MT rows this workgroup owns.  More than 16 rows (round 4: up to 64 utterances per
    // decoding step) are split over workgroups rather than looped over inside one: a workgroup that owns ALL rows reads
    // all of A through one CU's load path (~30 B / clk measured: fc2 at 64 rows 19.3 us against 7.7 us at 16,
    // profiles/r04_skinny_bench.txt), while its 16 weight columns are an L2 hit for the other row blocks anyway.
    const int mb = blockIdx.y * 16 * MT;
    A += (long)mb * lda;
    C += (long)mb * ldc;
    if (resid) resid += (long)mb * ldr;
    M -= mb;
  }
  const int m_glob0 = blockIdx.y * 16 * MT;
  // row_map (round 6: several batches decoded as ONE merged batch): the dropout mask of row m is the one row row_map[m] of its
  // OWN batch would draw -- the merged decode returns the hypotheses of one batch after the other, bit for bit
  const int m_last = m_glob0 + M - 1;
  auto drow = [&](int m) -> uint64_t { return (uint64_t)(row_map ? row_map[min(m_glob0 + m, m_last)] : m_glob0 + m); };
  if (ln_g) {
    // fused LayerNorm of the activation rows (the decoder's pre-LN in front of a projection): every workgroup
    // recomputes the row statistics (L2-resident input) instead of a separate kernel + round trip.
    // 16 threads per row, two passes (mean, then squared deviations) like layernorm_fwd_kernel.
    // (the MT row blocks advance together: their loads are independent, so a pass costs one memory round trip, not MT)
    const int j = tid & 15;
    const float* xr[MT];
    float s[MT], q[MT], mean[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      xr[mt] = A + (long)min(16 * mt + (tid >> 4), M - 1) * lda;
      s[mt] = q[mt] = 0.f;
    }
    // (UL chunks of 64 columns per trip, all their loads issued before the first add: a pass is K / (64 UL) memory round
    // trips -- two at K = 512 -- instead of K / 64; the summation order over c is unchanged)
    constexpr int UL = 8;
    for (int c0 = 4 * j; c0 < K; c0 += 64 * UL) {
      float4 v[UL][MT];
#pragma unroll
      for (int u = 0; u < UL; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) v[u][mt] = *reinterpret_cast<const float4*>(xr[mt] + min(c0 + 64 * u, K - 4));
#pragma unroll
      for (int u = 0; u < UL; ++u)
        if (c0 + 64 * u < K)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) s[mt] += v[u][mt].x + v[u][mt].y + v[u][mt].z + v[u][mt].w;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float t = s[mt];
      t += __shfl_xor(t, 8); t += __shfl_xor(t, 4); t += __shfl_xor(t, 2); t += __shfl_xor(t, 1);
      mean[mt] = t / K;
    }
    for (int c0 = 4 * j; c0 < K; c0 += 64 * UL) {
      float4 v[UL][MT];
#pragma unroll
      for (int u = 0; u < UL; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) v[u][mt] = *reinterpret_cast<const float4*>(xr[mt] + min(c0 + 64 * u, K - 4));
#pragma unroll
      for (int u = 0; u < UL; ++u)
        if (c0 + 64 * u < K)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float a = v[u][mt].x - mean[mt], b = v[u][mt].y - mean[mt], cc = v[u][mt].z - mean[mt], d = v[u][mt].w - mean[mt];
            q[mt] += a * a + b * b + cc * cc + d * d;
          }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float t = q[mt];
      t += __shfl_xor(t, 8); t += __shfl_xor(t, 4); t += __shfl_xor(t, 2); t += __shfl_xor(t, 1);
      if (j == 0) { ln_mean[16 * mt + (tid >> 4)] = mean[mt]; ln_rstd[16 * mt + (tid >> 4)] = rsqrtf(t / K + ln_eps); }
    }
    __syncthreads();
  }
  float mu[MT], rs[MT];
  const float* arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    mu[mt] = ln_g ? ln_mean[16 * mt + (lane & 15)] : 0.f;
    rs[mt] = ln_g ? ln_rstd[16 * mt + (lane & 15)] : 1.f;
    arow[mt] = A + (long)min(16 * mt + (lane & 15), M - 1) * lda + kc;
  }
  const bf16_t* wrow = W + (long)min(n0 + (lane & 15), N - 1) * ldw + kc;
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int steps = K >> 5;
  constexpr int U = 4;  // K-steps in flight per wave and pass (2 float4 per row block and step; a decoding launch has at most
                        // one workgroup per CU, so the 512-register budget of one wave per SIMD is there to be used)
  for (int s0 = wave; s0 < steps; s0 += 4 * U) {
    float4 a0[U][MT], a1[U][MT], lg0[U], lg1[U], lb0[U], lb1[U];
    uint4 w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = min(s0 + 4 * u, steps - 1);  // clamped: loads stay unconditional (issued back to back)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        a0[u][mt] = *reinterpret_cast<const float4*>(arow[mt] + 32 * s);
        a1[u][mt] = *reinterpret_cast<const float4*>(arow[mt] + 32 * s + 4);
      }
      w[u] = *reinterpret_cast<const uint4*>(wrow + 32 * s);
      if (ln_g) {  // (the step's gamma / beta chunks ride with its operand loads)
        const int k = 32 * s + kc;
        lg0[u] = *reinterpret_cast<const float4*>(ln_g + k); lg1[u] = *reinterpret_cast<const float4*>(ln_g + k + 4);
        lb0[u] = *reinterpret_cast<const float4*>(ln_b + k); lb1[u] = *reinterpret_cast<const float4*>(ln_b + k + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (s0 + 4 * u >= steps) break;
      union { uint4 q; bf16x8 v; } xa, wb;
      wb.q = w[u];
      const float4 g0 = lg0[u], g1 = lg1[u], b0 = lb0[u], b1 = lb1[u];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float4 p0 = a0[u][mt], p1 = a1[u][mt];
        if (ln_g) {
          const float m_ = mu[mt], r_ = rs[mt];
          p0.x = (p0.x - m_) * r_ * g0.x + b0.x; p0.y = (p0.y - m_) * r_ * g0.y + b0.y;
          p0.z = (p0.z - m_) * r_ * g0.z + b0.z; p0.w = (p0.w - m_) * r_ * g0.w + b0.w;
          p1.x = (p1.x - m_) * r_ * g1.x + b1.x; p1.y = (p1.y - m_) * r_ * g1.y + b1.y;
          p1.z = (p1.z - m_) * r_ * g1.z + b1.z; p1.w = (p1.w - m_) * r_ * g1.w + b1.w;
        }
        const uint2 lo = pack_bf16x4(p0.x, p0.y, p0.z, p0.w), hi = pack_bf16x4(p1.x, p1.y, p1.z, p1.w);
        xa.q = make_uint4(lo.x, lo.y, hi.x, hi.y);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb.v, xa.v, acc[mt], 0, 0, 0);
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4*>(&part[wave - 1][(mt * 64 + lane) * 4]) = acc[mt];
  }
  __syncthreads();
  if (wave > 0) return;
  const int n = n0 + (lane >> 4) * 4;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  // Four whole columns per lane and 16-byte aligned rows (every product of a decoding step but the 1-wide stop head): the
  // bias and ALL row blocks' residual rows are fetched as float4 before the first use.  The first form loaded them
  // element by element behind the `n + r < N` test: 4 MT dependent round trips, ~0.55 us each -- the whole difference
  // between 16 and 64 rows (out-proj + residual 4.4 -> 11.1 us, profiles/r04_skinny_bench.txt).
  const bool vec4 = n + 3 < N && (ldc & 3) == 0 && (ldr & 3) == 0 && ((uintptr_t)C & 15) == 0 &&
                    (!resid || ((uintptr_t)resid & 15) == 0) && (!bias || ((uintptr_t)bias & 15) == 0) && (n0 & 3) == 0;
  if (vec4) {
    float4 bv = {0.f, 0.f, 0.f, 0.f}, rv[MT];
    if (bias) bv = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = min(16 * mt + (lane & 15), M - 1);
      rv[mt] = resid ? *reinterpret_cast<const float4*>(resid + (long)m * ldr + n) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(&part[w][(mt * 64 + lane) * 4]);
        acc[mt][0] += p[0]; acc[mt][1] += p[1]; acc[mt][2] += p[2]; acc[mt][3] += p[3];
      }
      const int m = 16 * mt + (lane & 15);
      if (m >= M) continue;
      float v[4] = {acc[mt][0] + bv.x, acc[mt][1] + bv.y, acc[mt][2] + bv.z, acc[mt][3] + bv.w};
      const float rr[4] = {rv[mt].x, rv[mt].y, rv[mt].z, rv[mt].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = v[r];
        if (act == 1) x = fmaxf(x, 0.f);
        else if (act == 2) x = gelu_erf(x);
        else if (act == 3) x = 1.f / (1.f + __expf(-x));
        if (drop_p > 0.f) x *= drop_scale(seed, drow(m) * (uint64_t)N + n + r, drop_p, inv_keep);
        v[r] = x + rr[r];
      }
      *reinterpret_cast<float4*>(C + (long)m * ldc + n) = float4{v[0], v[1], v[2], v[3]};
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(&part[w][(mt * 64 + lane) * 4]);
      acc[mt][0] += p[0]; acc[mt][1] += p[1]; acc[mt][2] += p[2]; acc[mt][3] += p[3];
    }
    const int m = 16 * mt + (lane & 15);
    if (m >= M || n >= N) continue;
    float v[4] = {acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r >= N) break;
      float x = v[r];
      if (bias) x += bias[n + r];
      if (act == 1) x = fmaxf(x, 0.f);
      else if (act == 2) x = gelu_erf(x);
      else if (act == 3) x = 1.f / (1.f + __expf(-x));
      if (drop_p > 0.f) x *= drop_scale(seed, drow(m) * (uint64_t)N + n + r, drop_p, inv_keep);
      if (resid) x += resid[(long)m * ldr + n + r];
      C[(long)m * ldc + n + r] = x;
    }
  }
}

// A fp32 [M][K] (row stride lda), W bf16 [N][K] (row stride ldw): M <= 64, K % 32 == 0, 16-byte aligned rows
// ln_g / ln_b (optional, K floats each): y = f(LayerNorm(x) W^T + b) -- the normalisation is applied to the rows
// while they are converted (K % 64 == 0 then).  resid rows may all be the same one (ldr = 0: a positional row).
int s2st_gemm_skinny(const float* A, long lda, const bf16raw* W, long ldw, float* C, long ldc, const float* bias, int act,
                     float drop_p, uint64_t seed, const float* resid, long ldr, int M, int N, int K, hipStream_t st,
                     const float* ln_g, const float* ln_b, float ln_eps, const uint64_t* seed_ptr, const int* row_map) {
  if (M <= 0 || N <= 0) return 0;
  if (M > S2ST_SKINNY_MAX_ROWS || K <= 0 || K % 32 || lda % 4 || ldw % 8 || ((uintptr_t)A % 16) || ((uintptr_t)W % 16)) return S2ST_ERR_SHAPE;
  if (ln_g && (!ln_b || K % 64 || ((uintptr_t)ln_g % 16) || ((uintptr_t)ln_b % 16))) return S2ST_ERR_SHAPE;
  // bytes the launch has to move: the bf16 weight rows once, the fp32 activation rows, the result (+ residual)
  const double by = 2.0 * N * K + 4.0 * M * K + 4.0 * M * N * (resid ? 2 : 1);
  const double fl = 2.0 * M * N * (double)K;
  const dim3 grid((N + 15) / 16, (M + 15) / 16), block(256);  // (16 columns) x (16 rows) per workgroup
  const bf16_t* Wp = reinterpret_cast<const bf16_t*>(W);
  s2st_launch("gemm_skinny_kernel", by, fl, gemm_skinny_kernel<1>, grid, block, 0, st, A, lda, Wp, ldw, C, ldc, bias, act, drop_p,
              seed, resid, ldr, M, N, K, ln_g, ln_b, ln_eps, seed_ptr, row_map);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
