// General strided / batched GEMM for gfx950:  C(m,n) = epi(alpha * sum_k A(m,k) * B(n,k)).
//
// Replaces (on the hot path) every F.linear / F.conv1d / torch.bmm of the reference:
// fairseq/modules/multihead_attention.py:170-192, 332, 367; transformer_layer.py:158-162;
// examples/s2s_trans/models/s2st_transformer.py:135-139, 452-455 (convs run as GEMMs over
// halo-padded [B][T+2h][C] buffers: row (b,t) of the im2col matrix is the contiguous
// window starting at xpad[b][t*stride], so no im2col is materialised).
//
// Structure: 256 threads = 4 waves (2x2), block tile BMxBN (128x128 or 64x64), BK = 32.
// fp32 operands are loaded with 16-byte global loads, converted to bf16 in registers
// (v_cvt_pk_bf16_f32) and written K-contiguous into a double-buffered, padded LDS image;
// "rows-contiguous" operands (weight-gradient and P*V forms) are transposed 4x4 in
// registers on the way.  Fragments are read with ds_read_b128 and fed to
// v_mfma_f32_16x16x32_bf16; global loads of tile t+1 are in flight during the MFMAs of
// tile t (one barrier per K-step).  Epilogue fuses alpha, bias, ReLU, dropout, residual,
// accumulate / split-K atomics.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "s2st_ops.h"
#include "s2st_prof.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_ROW = 80;  // bytes per tile row: 32 bf16 (64 B) + 16 B pad

// LDS images (bf16):
//   K-contiguous operand  : [row][32 k], pitch LDS_ROW (80 B); fragments by ds_read_b128.
//   rows-contiguous operand: natural [k slot][rows], pitch 2*ROWS + 32 B, k rows stored at slot
//     s(k) = k with bits 2 and 3 swapped so that the 8 k rows one 32-lane half reads with
//     ds_read_b64_tr_b16 ({0-3, 8-11} + 4h) are adjacent 32-B chunks: conflict-free reads, and
//     the stores (consecutive lanes = consecutive row quads of one k) are contiguous.
__device__ __forceinline__ constexpr int kslot(int k) {
  return (k & 19) | ((k & 4) << 1) | ((k & 8) >> 1);
}

template <bool KM, int ROWS, bool VEC>
struct TileLoader {
  // KM  : slot i -> f = tid + 256 i ; row = f >> 3 ; kq = f & 7 (= tid & 7)
  // !KM : slot i -> f = tid + 256 i ; rq = f % (ROWS/4) ; kk = f / (ROWS/4)
  // VEC : every access is an aligned float4 and the loop body is branch-free: out-of-range
  //       rows are CLAMPED (their products land in accumulator rows/cols the epilogue never
  //       stores), the K tail is clamped and zeroed by a select.  !VEC: guarded scalar loads.
  static constexpr int NLD = ROWS * 8 / 256;
  static constexpr int RQ = ROWS / 4;
  static constexpr int PITCH = KM ? LDS_ROW : (2 * ROWS + 32);
  static constexpr int BYTES = KM ? ROWS * LDS_ROW : BK * (2 * ROWS + 32);
  float r[NLD][4];
  long off[NLD];  // KM: row offset (VEC: clamped row; !VEC: -1 = out of range)
  int tid, row0;  // !KM: first of this thread's 4 rows (VEC: clamped)

  __device__ __forceinline__ void init(const GemmOperand& X, int r0, int R, int tid_) {
    tid = tid_;
    if (KM) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        int row = r0 + ((tid + 256 * i) >> 3);
        if (VEC) off[i] = split_off(X.sp, min(row, R - 1));
        else off[i] = row < R ? split_off(X.sp, row) : -1;
      }
    } else {
      row0 = r0 + 4 * (tid % RQ);
      if (VEC) row0 = min(row0, R - 4);
    }
  }

  __device__ __forceinline__ void load(const GemmOperand& X, const float* base, int R, int kt, int kend) {
    if (KM) {
      const int k = kt + (tid & 7) * 4;
      if (VEC) {
        const bool ok = k < kend;
        const int kc = min(k, kend - 4);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          float4 v = *reinterpret_cast<const float4*>(base + off[i] + kc);
          r[i][0] = ok ? v.x : 0.f; r[i][1] = ok ? v.y : 0.f;
          r[i][2] = ok ? v.z : 0.f; r[i][3] = ok ? v.w : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            r[i][e] = (off[i] >= 0 && k + e < kend) ? base[off[i] + k + e] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int k = kt + (tid + 256 * i) / RQ;
        const bool ok = k < kend;
        const long o = split_off(X.sp, min(k, kend - 1));
        if (VEC) {
          float4 v = *reinterpret_cast<const float4*>(base + o + row0);
          r[i][0] = ok ? v.x : 0.f; r[i][1] = ok ? v.y : 0.f;
          r[i][2] = ok ? v.z : 0.f; r[i][3] = ok ? v.w : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) r[i][e] = (ok && row0 + e < R) ? base[o + row0 + e] : 0.f;
        }
      }
    }
  }

  template <bool PRECISE>
  __device__ __forceinline__ void store(unsigned char* hi_img, unsigned char* lo_img) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + 256 * i;
      const int o = KM ? (f >> 3) * LDS_ROW + (f & 7) * 8 : kslot(f / RQ) * PITCH + (f % RQ) * 8;
      if (PRECISE) {
        uint2 h, l;
        split_bf16x4(r[i][0], r[i][1], r[i][2], r[i][3], h, l);
        *reinterpret_cast<uint2*>(hi_img + o) = h;
        *reinterpret_cast<uint2*>(lo_img + o) = l;
      } else {
        *reinterpret_cast<uint2*>(hi_img + o) = pack_bf16x4(r[i][0], r[i][1], r[i][2], r[i][3]);
      }
    }
  }

  // MFMA 16x16x32 operand fragment for tile rows [rt, rt + 16): lane l holds k = 8 (l>>4) + j
  __device__ static __forceinline__ bf16x8 frag(const unsigned char* img, int rt, int lane) {
    if (KM) {
      return *reinterpret_cast<const bf16x8*>(img + (rt + (lane & 15)) * LDS_ROW + (lane >> 4) * 16);
    } else {
      const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
      const unsigned char* a = img + kslot(8 * g + q) * PITCH + (rt + 4 * p) * 2;
      s16x4 lo = lds_read_tr16(a);
      s16x4 hi = lds_read_tr16(a + (kslot(4) - kslot(0)) * PITCH);  // k rows +4 -> slot +8
      bf16x8 f;
      f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
      f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
      return f;
    }
  }
};

template <int BM, int BN, bool AKM, bool BKM, bool PRECISE, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
  typedef TileLoader<AKM, BM, VEC> LA;
  typedef TileLoader<BKM, BN, VEC> LB;
  constexpr int A_BYTES = LA::BYTES, B_BYTES = LB::BYTES;
  constexpr int IMG = PRECISE ? 2 : 1;
  constexpr int STAGE = (A_BYTES + B_BYTES) * IMG;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int zb = blockIdx.z / g.splitk, ks = blockIdx.z - zb * g.splitk;
  const int zq = zb / g.zdiv, zr = zb - zq * g.zdiv;
  const float* abase = reinterpret_cast<const float*>(g.A.p) + zq * g.A.zo + zr * g.A.zi;
  const float* bbase = reinterpret_cast<const float*>(g.B.p) + zq * g.B.zo + zr * g.B.zi;
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

  // stage layout: [A hi][B hi]([A lo][B lo])
  if (nt > 0) {
    la.load(g.A, abase, g.M, kbeg, kend);
    lb.load(g.B, bbase, g.N, kbeg, kend);
    la.template store<PRECISE>(smem, smem + A_BYTES + B_BYTES);
    lb.template store<PRECISE>(smem + A_BYTES, smem + A_BYTES + B_BYTES + A_BYTES);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    unsigned char* cur = smem + (t & 1) * STAGE;
    unsigned char* nxt = smem + ((t + 1) & 1) * STAGE;
    const bool more = t + 1 < nt;
    if (more) {
      la.load(g.A, abase, g.M, kbeg + (t + 1) * BK, kend);
      lb.load(g.B, bbase, g.N, kbeg + (t + 1) * BK, kend);
    }
    bf16x8 af[TM], bf[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = LA::frag(cur, wm * WM + i * 16, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = LB::frag(cur + A_BYTES, wn * WN + j * 16, lane);
    if (PRECISE) {
      bf16x8 al[TM], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) al[i] = LA::frag(cur + A_BYTES + B_BYTES, wm * WM + i * 16, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) bl[j] = LB::frag(cur + A_BYTES + B_BYTES + A_BYTES, wn * WN + j * 16, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bf[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bl[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    if (more) {
      la.template store<PRECISE>(nxt, nxt + A_BYTES + B_BYTES);
      lb.template store<PRECISE>(nxt + A_BYTES, nxt + A_BYTES + B_BYTES + A_BYTES);
    }
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------
  float* cbase = g.C.p + zq * g.C.zo + zr * g.C.zi;
  const float* rbase = g.ep.resid ? g.ep.resid + zq * g.C.zo + zr * g.C.zi : nullptr;
  const float inv_keep = g.ep.drop_p > 0.f ? 1.f / (1.f - g.ep.drop_p) : 1.f;
  const bool lead = (ks == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int m = m0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
      if (m >= g.M) continue;
      long roff = split_off(g.C.sp, m);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int n = n0 + wn * WN + j * 16 + (lane & 15);
        if (n >= g.N) continue;
        float v = g.ep.alpha * acc[i][j][r];
        if (g.ep.bias && lead) v += g.ep.bias[zq * g.ep.bias_zo + n];
        if (g.ep.act == 1) v = fmaxf(v, 0.f);
        else if (g.ep.act == 2) v = gelu_erf(v);
        if (g.ep.drop_p > 0.f)
          v *= drop_scale(g.ep.seed, ((uint64_t)zb * g.M + m) * (uint64_t)g.N + n, g.ep.drop_p, inv_keep);
        if (rbase && lead) v += rbase[roff + n];
        float* cp = cbase + roff + n;
        if (g.splitk > 1) {
          atomicAdd(cp, v);
        } else if (g.ep.accumulate) {
          *cp += v;
        } else {
          *cp = v;
        }
      }
    }
  }
}

template <int BM, int BN, bool PRECISE, bool VEC>
void launch_layouts(const GemmArgs& g, dim3 grid, hipStream_t st) {
  if (g.A.kmajor && g.B.kmajor)
    S2ST_LAUNCH((gemm_kernel<BM, BN, true, true, PRECISE, VEC>), grid, dim3(256), 0, st, g);
  else if (g.A.kmajor && !g.B.kmajor)
    S2ST_LAUNCH((gemm_kernel<BM, BN, true, false, PRECISE, VEC>), grid, dim3(256), 0, st, g);
  else if (!g.A.kmajor && g.B.kmajor)
    S2ST_LAUNCH((gemm_kernel<BM, BN, false, true, PRECISE, VEC>), grid, dim3(256), 0, st, g);
  else
    S2ST_LAUNCH((gemm_kernel<BM, BN, false, false, PRECISE, VEC>), grid, dim3(256), 0, st, g);
}

// aligned-float4 fast path: pointer/strides 16-byte aligned and the contiguous extent a multiple of 4
bool vec_ok(const GemmOperand& o, int R, int K) {
  bool base = ((uintptr_t)o.p % 16 == 0) && (o.sp.ld % 4 == 0) && (o.sp.bs % 4 == 0) &&
              (o.zo % 4 == 0) && (o.zi % 4 == 0) && K >= 4;
  return base && (o.kmajor ? (K % 4 == 0) : (R % 4 == 0 && R >= 4));
}

}  // namespace

// ---- per-dispatch timing registry (bench.py roofline leg; s2st_prof.h) -----------------------------------------
namespace {
struct ProfRec { const char* tag; hipEvent_t a, b; double work, work2; hipStream_t st; };
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
}  // namespace

bool s2st_prof_enabled() { return g_prof_on; }
void s2st_prof_push(const char* tag, hipEvent_t a, hipEvent_t b, double work, double work2, hipStream_t st) {
  g_prof.push_back(ProfRec{tag, a, b, work, work2, st});
}
void s2st_profile_enable_impl(int on) { g_prof_on = on != 0; }

// One text line per tag: "tag\tlaunches\ttotal_us\twork\twork2\n" (work = FLOPs or bytes summed over the launches);
// clears the registry.  Returns the number of bytes written (without the terminating 0) or -1 if `cap` is too small.
// Timeline form (mode 1): one line per dispatch in launch order, "tag\tstream\tstart_us\tdur_us\n" -- stream = index
// in order of first use, start relative to the first dispatch's start (GPU clock, across streams).
static long profile_timeline(char* out, long cap) {
  std::vector<hipStream_t> streams;
  long o = 0;
  for (auto& r : g_prof) hipEventSynchronize(r.b);
  for (auto& r : g_prof) {
    int si = -1;
    for (size_t i = 0; i < streams.size(); ++i) if (streams[i] == r.st) si = (int)i;
    if (si < 0) { streams.push_back(r.st); si = (int)streams.size() - 1; }
    float t0 = 0.f, e = 0.f;
    hipEventElapsedTime(&t0, g_prof[0].a, r.a);
    hipEventElapsedTime(&e, r.a, r.b);
    int w = snprintf(out + o, cap > o ? (size_t)(cap - o) : 0, "%s\t%d\t%.3f\t%.3f\n", r.tag, si, t0 * 1e3, e * 1e3);
    if (w < 0 || o + w >= cap) { o = -1; break; }
    o += w;
  }
  for (auto& r : g_prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  g_prof.clear();
  return o;
}

long s2st_profile_report_impl(char* out, long cap, int mode) {
  if (mode == 1) return profile_timeline(out, cap);
  struct Agg { const char* tag; long n; double us, work, work2; };
  std::vector<Agg> aggs;
  for (auto& r : g_prof) {
    hipEventSynchronize(r.b);
    float e = 0.f;
    hipEventElapsedTime(&e, r.a, r.b);
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
    Agg* f = nullptr;
    for (auto& a : aggs) if (strcmp(a.tag, r.tag) == 0) { f = &a; break; }
    if (!f) { aggs.push_back(Agg{r.tag, 0, 0, 0, 0}); f = &aggs.back(); }
    f->n += 1; f->us += e * 1e3; f->work += r.work; f->work2 += r.work2;
  }
  g_prof.clear();
  long o = 0;
  for (auto& a : aggs) {
    int w = snprintf(out + o, cap > o ? (size_t)(cap - o) : 0, "%s\t%ld\t%.3f\t%.6e\t%.6e\n", a.tag, a.n, a.us, a.work, a.work2);
    if (w < 0 || o + w >= cap) return -1;
    o += w;
  }
  return o;
}

int s2st_gemm(GemmArgs g, hipStream_t st, int* tile_out) {
  if (tile_out) *tile_out = 0;
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return 0;
  if (g.A.dtype != g.B.dtype) return S2ST_ERR_ARG;
  const bool bf16_in = g.A.dtype == S2ST_BF16;
  if (bf16_in && g.precise) return S2ST_ERR_ARG;       // bf16x3 needs the fp32 values
  if (!bf16_in && (g.C.h || !g.C.p)) return S2ST_ERR_ARG;  // bf16 copy: fast path only
  if (g.zdiv <= 0) g.zdiv = 1;
  if (bf16_in) {
    int tile = 0;
    const int rc = s2st_gemm_bf16(g, st, &tile);
    if (tile_out) *tile_out = tile;
    return rc;
  }
  g.avec = vec_ok(g.A, g.M, g.K) ? 1 : 0;
  g.bvec = vec_ok(g.B, g.N, g.K) ? 1 : 0;
  const bool vec = g.avec && g.bvec;
  auto tiles = [&](int bm, int bn) {
    return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * g.batch;
  };
  bool linear_epi = !g.ep.act && g.ep.drop_p == 0.f;
  bool wgrad_like = g.ep.accumulate && linear_epi && g.K >= 16 * BK;
  bool big = vec && !g.precise && (tiles(128, 128) >= 192 || (wgrad_like && g.M >= 128 && g.N >= 128));
  int bm = big ? 128 : 64, bn = bm;
  long nt = tiles(bm, bn);
  // split-K only for accumulating outputs (weight gradients): partial sums are added with
  // fp32 atomics, the epilogue must then be linear.
  g.splitk = 1;
  if (g.ep.accumulate && linear_epi && nt < 256 && g.K >= 8 * BK) {
    int want = (int)((512 + nt - 1) / nt);
    int maxs = g.K / (4 * BK);
    g.splitk = want < maxs ? want : maxs;
    if (g.splitk < 1) g.splitk = 1;
  }
  int kt = (g.K + BK - 1) / BK;
  g.kchunk = ((kt + g.splitk - 1) / g.splitk) * BK;  // multiple of 32: split chunks keep K % 4
  g.splitk = (g.K + g.kchunk - 1) / g.kchunk;
  if (g.splitk < 1) g.splitk = 1;
  dim3 grid((g.N + bn - 1) / bn, (g.M + bm - 1) / bm, g.batch * g.splitk);
  if (grid.y > 65535 || grid.z > 65535) return -2;
  if (g.precise) {
    if (vec) launch_layouts<64, 64, true, true>(g, grid, st);
    else launch_layouts<64, 64, true, false>(g, grid, st);
  } else if (big) {
    launch_layouts<128, 128, false, true>(g, grid, st);
  } else {
    if (vec) launch_layouts<64, 64, false, true>(g, grid, st);
    else launch_layouts<64, 64, false, false>(g, grid, st);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
