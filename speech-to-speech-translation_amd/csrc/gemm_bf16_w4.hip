// bf16 GEMM, "early-release" ring form for SHORT-K products (round 3): C(m, n) = epi(alpha * sum_k A(m, k) B(n, k)).
//
// Why a second form.  The 8-wave ring kernel of gemm_bf16.hip holds 128 KB of LDS, i.e. ONE workgroup per CU, and the
// products of a training step are short (K = 512 ... 2048: 8 ... 32 K-steps): in-kernel stamps (profiles/r02_gemm_stamps.txt)
// put 3.0 k cycles of prologue and 2.2 - 5.4 k cycles of epilogue next to a 7.6 k-cycle K-loop, with nothing on the CU
// to overlap them with.  Here a workgroup is 4 waves (2 x 2, one per SIMD, wave tile 64 x 64 for a 128 x 128 block
// tile: every LDS byte of a stage is read by two waves instead of four) and its ring is small enough for TWO (128 x
// 128) or THREE (128 x 64) workgroups per CU, so one tile's prologue / epilogue / barrier waits run under another
// tile's MFMAs -- the SIMD's two (three) resident waves belong to different tiles and are not phase-locked by a
// barrier.
//
// The small ring is kept deep by releasing a slot EARLY: the fragments of K-step t go LDS -> registers (64 VGPRs for
// a 64 x 64 wave tile) at the START of the step, a second barrier certifies that every wave has them, and the DMA of
// step t + NS is issued into the slot just read BEFORE the step's 32 MFMAs -- NS K-steps of DMA in flight from an
// NS-slot ring (the classic ring issues step t + NS - 1 into the slot of step t - 1: NS - 1 in flight).
//   step t:  vmcnt (own DMA pieces of step t landed) | barrier A (everyone's pieces) | fragment reads, lgkmcnt(0) |
//            barrier B (slot t % NS is free) | issue DMA of step t + NS | MFMAs of step t
// Tile images, swizzles, fragment readers, K-tail handling and epilogues are the ones of gemm_bf16.hip
// (gemm_bf16_tile.h).  Replaces F.linear of fairseq/modules/transformer_layer.py:140-162 and
// multihead_attention.py:170-192 (projections) in fast mode, like the kernels it sits beside.
#include "gemm_bf16_tile.h"

namespace {

#define S2ST_LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f)  // lgkmcnt(0), vmcnt / expcnt untouched

template <int BM, int BN, bool AKM, bool BKM, int NS>
__device__ __forceinline__ void gemm_w4_tile(const GemmArgs& g, int id, const int nwg, const int by) {
  constexpr int NW = 4, WGN = 2;
  constexpr int WM = BM / 2, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  typedef Dma<AKM, BM, NW> DA;
  typedef Dma<BKM, BN, NW> DB;
  typedef Stage<AKM, BM, true> LA;
  typedef Stage<BKM, BN, true> LB;
  constexpr int A_BYTES = DA::BYTES, B_BYTES = DB::BYTES, STAGE = A_BYTES + B_BYTES;
  constexpr int PER_STAGE = DA::NI + DB::NI;
  static_assert(NS >= 2 && NS <= 4, "ring depth");
  static_assert(NS * PER_STAGE < 64, "vmcnt field");
  HIP_DYNAMIC_SHARED(unsigned char, smem)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  {  // XCD-aware tile order: ids that share an XCD (id % 8) own a contiguous run of tiles
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

  DA da;
  DB db;
  da.init(g.A, abase, m0, g.M, wave, lane);
  db.init(g.B, bbase, n0, g.N, wave, lane);

  // prologue: stages 0 .. NS-1 (the whole ring)
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (s < nt) {
      da.issue(smem + s * STAGE, kbeg + s * BK, g.K, wave);
      db.issue(smem + s * STAGE + A_BYTES, kbeg + s * BK, g.K, wave);
    }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int slot = 0;
  for (int t = 0; t < nt; ++t) {
    // my DMA pieces of step t have landed once at most min(NS - 1, nt - 1 - t) younger stages are outstanding
    const int ahead = nt - 1 - t;
    if (ahead >= NS - 1) S2ST_VMCNT((NS - 1) * PER_STAGE);
    else if (NS > 3 && ahead == 2) S2ST_VMCNT(2 * PER_STAGE);
    else if (NS > 2 && ahead == 1) S2ST_VMCNT(PER_STAGE);
    else S2ST_VMCNT(0);
    __builtin_amdgcn_s_barrier();  // A: every wave's pieces of step t are in LDS
    unsigned char* cur = smem + slot * STAGE;
    if (t == nt - 1 && kend - (kbeg + t * BK) < BK) {  // K tail: zero the invalid k of this stage
      const int kv = kend - (kbeg + t * BK);
      DA::sanitize(cur, kv, tid);
      DB::sanitize(cur + A_BYTES, kv, tid);
      __syncthreads();
    }
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
      S2ST_LGKM0();  // plain and hand-issued reads alike: the slot is about to be handed back
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[s][j] = LB::done(br[s][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[s][i] = LA::done(ar[s][i]);
      }
    }
    if (t + NS < nt) {  // (wave-uniform and the same in every wave: the barrier is met by all)
      __builtin_amdgcn_s_barrier();  // B: every wave holds its fragments of step t -- the slot is free
      da.issue(cur, kbeg + (t + NS) * BK, g.K, wave);
      db.issue(cur + A_BYTES, kbeg + (t + NS) * BK, g.K, wave);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
    slot = slot + 1 == NS ? 0 : slot + 1;
  }
  gemm_epilogue<BM, BN, WGN, true>(g, acc, m0, n0, wm, wn, lane, zb, ks, zq, zr);
}

// workgroups per CU the ring leaves room for (160 KB of LDS): the register budget follows from it
template <int BM, int BN, int NS>
constexpr int w4_wgs_per_cu() { return (160 * 1024) / (NS * (BM + BN) * 128) >= 3 ? 3 : ((160 * 1024) / (NS * (BM + BN) * 128) >= 2 ? 2 : 1); }

template <int BM, int BN, bool AKM, bool BKM, int NS>
__global__ __launch_bounds__(256, (w4_wgs_per_cu<BM, BN, NS>())) void gemm_bf16_w4_kernel(GemmArgs g) {
  gemm_w4_tile<BM, BN, AKM, BKM, NS>(g, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

// grouped one-shot form (a layer's weight-gradient products: the concatenated tile list, one workgroup per tile)
template <int BM, int BN, bool AKM, bool BKM, int NS>
__global__ __launch_bounds__(256, (w4_wgs_per_cu<BM, BN, NS>())) void gemm_bf16_w4_group_kernel(GemmGroup grp) {
  int pi = 0;
  const int t = blockIdx.x;
  while (pi + 1 < grp.n && t >= grp.tile0[pi + 1]) ++pi;
  gemm_w4_tile<BM, BN, AKM, BKM, NS>(grp.g[pi], t - grp.tile0[pi], grp.tile0[pi + 1] - grp.tile0[pi], 0);
}

double w4_flops(const GemmArgs& g) { return 2.0 * g.M * g.N * (double)g.K * g.batch; }
double w4_min_bytes(const GemmArgs& g) {
  const double mn = (double)g.M * g.N * g.batch;
  return 2.0 * g.batch * ((double)g.M * g.K + (double)g.N * g.K) + mn * ((g.C.p ? 4 : 0) + (g.C.h ? 2 : 0)) +
         mn * 4 * ((g.ep.accumulate ? 1 : 0) + (g.ep.resid ? 1 : 0));
}

template <int BM, int BN, int NS, bool GROUP, class ARG>
int launch_w4(const ARG& arg, bool akm, bool bkm, dim3 grid, double fl, double by, hipStream_t st) {
  constexpr int LDS = NS * (BM + BN) * 128;
  static char tags[4][96];
  auto go = [&](auto kern, int ti) {
    static bool configured = false;  // one flag per instantiation (the lambda's operator() template)
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
        return -1;
      configured = true;
    }
    if (!tags[ti][0])
      snprintf(tags[ti], sizeof tags[ti], "gemm_bf16_w4%s_kernel<%d, %d, %s, %s, %d>", GROUP ? "_group" : "", BM, BN,
               (ti & 2) ? "true" : "false", (ti & 1) ? "true" : "false", NS);
    s2st_launch(tags[ti], fl, by, kern, grid, dim3(256), LDS, st, arg);
    return 0;
  };
  if constexpr (GROUP) {
    if (akm && bkm) return go(gemm_bf16_w4_group_kernel<BM, BN, true, true, NS>, 3);
    if (akm && !bkm) return go(gemm_bf16_w4_group_kernel<BM, BN, true, false, NS>, 2);
    if (!akm && bkm) return go(gemm_bf16_w4_group_kernel<BM, BN, false, true, NS>, 1);
    return go(gemm_bf16_w4_group_kernel<BM, BN, false, false, NS>, 0);
  } else {
    if (akm && bkm) return go(gemm_bf16_w4_kernel<BM, BN, true, true, NS>, 3);
    if (akm && !bkm) return go(gemm_bf16_w4_kernel<BM, BN, true, false, NS>, 2);
    if (!akm && bkm) return go(gemm_bf16_w4_kernel<BM, BN, false, true, NS>, 1);
    return go(gemm_bf16_w4_kernel<BM, BN, false, false, NS>, 0);
  }
}

}  // namespace

// g: prepared by s2st_gemm_bf16 (alignment flags, tiles_n for the chosen tile, kchunk / splitk, epilogue marks)
int s2st_gemm_bf16_w4(const GemmArgs& g, int bm, int bn, dim3 grid, hipStream_t st) {
  const bool akm = g.A.kmajor != 0, bkm = g.B.kmajor != 0;
  const double fl = w4_flops(g), by = w4_min_bytes(g);
  if (bm == 128 && bn == 128) return launch_w4<128, 128, 2, false>(g, akm, bkm, grid, fl, by, st);
  if (bm == 128 && bn == 64) {
    // two ring slots: 48 KB, three workgroups per CU (what w4_pick's slot count assumes; 4584 x 2048 x 512: 18.3 us against
    // 22.0 us with three slots = two workgroups per CU, measured in round 3)
    return launch_w4<128, 64, 2, false>(g, akm, bkm, grid, fl, by, st);
  }
  return S2ST_ERR_ARG;
}

int s2st_gemm_bf16_w4_group(const GemmGroup& grp, hipStream_t st) {
  double fl = 0, by = 0;
  for (int i = 0; i < grp.n; ++i) { fl += w4_flops(grp.g[i]); by += w4_min_bytes(grp.g[i]); }
  const bool akm = grp.g[0].A.kmajor != 0, bkm = grp.g[0].B.kmajor != 0;
  return launch_w4<128, 128, 2, true>(grp, akm, bkm, dim3(grp.total < 1 ? 1 : grp.total), fl, by, st);
}

int s2st_gemm_bf16_w4_preload(hipStream_t st) {
  GemmArgs g{};
  g.A.dtype = g.B.dtype = S2ST_BF16;
  g.splitk = 1; g.zdiv = 1; g.tiles_n = 1; g.batch = 1; g.kchunk = BK;
  int rc = 0;
  for (int lay = 0; lay < 4; ++lay) {
    g.A.kmajor = lay & 1; g.B.kmajor = (lay >> 1) & 1;
    rc |= s2st_gemm_bf16_w4(g, 128, 128, dim3(1), st);
    rc |= launch_w4<128, 64, 2, false>(g, g.A.kmajor != 0, g.B.kmajor != 0, dim3(1), 0.0, 0.0, st);
    GemmGroup grp{};
    grp.n = 1; grp.g[0] = g; grp.total = 0;
    rc |= s2st_gemm_bf16_w4_group(grp, st);
  }
  return rc || hipGetLastError() != hipSuccess ? -1 : 0;
}
