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
#include <vector>

#include "s2st_ops.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_ROW = 80;  // bytes per tile row: 32 bf16 (64 B) + 16 B pad

template <bool KM, int ROWS>
struct TileLoader {
  // KM  : slot i -> f = tid + 256 i ; row = f >> 3 ; kq = f & 7      (ROWS*8/256 slots)
  // !KM : rq = tid % (ROWS/4), kq = tid / (ROWS/4) (active if kq < 8) ; slot i -> k = 4 kq + i
  static constexpr int NLD = KM ? (ROWS * 8 / 256) : 4;
  float r[NLD][4];
  long off[NLD];   // KM: per-slot row offset (-1 = row out of range)
  int tid;

  __device__ __forceinline__ void init(const GemmOperand& X, int r0, int R, int tid_) {
    tid = tid_;
    if (KM) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        int row = r0 + ((tid + 256 * i) >> 3);
        off[i] = row < R ? split_off(X.sp, row) : -1;
      }
    }
  }

  __device__ __forceinline__ void load(const GemmOperand& X, const float* base, int r0, int R,
                                       int kt, int kend, int vec) {
    if (KM) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        int k = kt + ((tid + 256 * i) & 7) * 4;
        r[i][0] = r[i][1] = r[i][2] = r[i][3] = 0.f;
        if (off[i] >= 0 && k < kend) {
          const float* p = base + off[i] + k;
          if (vec && k + 3 < kend) {
            float4 v = *reinterpret_cast<const float4*>(p);
            r[i][0] = v.x; r[i][1] = v.y; r[i][2] = v.z; r[i][3] = v.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (k + e < kend) r[i][e] = p[e];
          }
        }
      }
    } else {
      constexpr int RQ = ROWS / 4;
      int rq = tid % RQ, kq = tid / RQ;
      int row = r0 + 4 * rq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        r[i][0] = r[i][1] = r[i][2] = r[i][3] = 0.f;
        int k = kt + 4 * kq + i;
        if (kq < 8 && k < kend && row < R) {
          const float* p = base + split_off(X.sp, k) + row;
          if (vec && row + 3 < R) {
            float4 v = *reinterpret_cast<const float4*>(p);
            r[i][0] = v.x; r[i][1] = v.y; r[i][2] = v.z; r[i][3] = v.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (row + e < R) r[i][e] = p[e];
          }
        }
      }
    }
  }

  template <bool PRECISE>
  __device__ __forceinline__ void store(unsigned char* hi_img, unsigned char* lo_img) {
    if (KM) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        int f = tid + 256 * i;
        int o = (f >> 3) * LDS_ROW + (f & 7) * 8;
        if (PRECISE) {
          uint2 h, l;
          split_bf16x4(r[i][0], r[i][1], r[i][2], r[i][3], h, l);
          *reinterpret_cast<uint2*>(hi_img + o) = h;
          *reinterpret_cast<uint2*>(lo_img + o) = l;
        } else {
          *reinterpret_cast<uint2*>(hi_img + o) = pack_bf16x4(r[i][0], r[i][1], r[i][2], r[i][3]);
        }
      }
    } else {
      constexpr int RQ = ROWS / 4;
      int rq = tid % RQ, kq = tid / RQ;
      if (kq < 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int o = (4 * rq + j) * LDS_ROW + kq * 8;
          if (PRECISE) {
            uint2 h, l;
            split_bf16x4(r[0][j], r[1][j], r[2][j], r[3][j], h, l);
            *reinterpret_cast<uint2*>(hi_img + o) = h;
            *reinterpret_cast<uint2*>(lo_img + o) = l;
          } else {
            *reinterpret_cast<uint2*>(hi_img + o) = pack_bf16x4(r[0][j], r[1][j], r[2][j], r[3][j]);
          }
        }
      }
    }
  }
};

template <int BM, int BN, bool AKM, bool BKM, bool PRECISE>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
  constexpr int A_BYTES = BM * LDS_ROW, B_BYTES = BN * LDS_ROW;
  constexpr int IMG = PRECISE ? 2 : 1;
  constexpr int STAGE = (A_BYTES + B_BYTES) * IMG;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int zb = blockIdx.z / g.splitk, ks = blockIdx.z - zb * g.splitk;
  const int zq = zb / g.zdiv, zr = zb - zq * g.zdiv;
  const float* abase = g.A.p + zq * g.A.zo + zr * g.A.zi;
  const float* bbase = g.B.p + zq * g.B.zo + zr * g.B.zi;
  const int kbeg = ks * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nt = (kend - kbeg + BK - 1) / BK;

  TileLoader<AKM, BM> la;
  TileLoader<BKM, BN> lb;
  la.init(g.A, m0, g.M, tid);
  lb.init(g.B, n0, g.N, tid);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nt > 0) {
    la.load(g.A, abase, m0, g.M, kbeg, kend, g.avec);
    lb.load(g.B, bbase, n0, g.N, kbeg, kend, g.bvec);
    la.template store<PRECISE>(smem, smem + A_BYTES + B_BYTES);
    lb.template store<PRECISE>(smem + A_BYTES, smem + A_BYTES + B_BYTES + A_BYTES);
  }
  __syncthreads();

  const int frag_off = (lane & 15) * LDS_ROW + (lane >> 4) * 16;
  for (int t = 0; t < nt; ++t) {
    unsigned char* cur = smem + (t & 1) * STAGE;
    unsigned char* nxt = smem + ((t + 1) & 1) * STAGE;
    const bool more = t + 1 < nt;
    if (more) {
      la.load(g.A, abase, m0, g.M, kbeg + (t + 1) * BK, kend, g.avec);
      lb.load(g.B, bbase, n0, g.N, kbeg + (t + 1) * BK, kend, g.bvec);
    }
    bf16x8 af[TM], bf[TN];
    const unsigned char* As = cur + wm * WM * LDS_ROW + frag_off;
    const unsigned char* Bs = cur + A_BYTES + wn * WN * LDS_ROW + frag_off;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + i * 16 * LDS_ROW);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 16 * LDS_ROW);
    if (PRECISE) {
      bf16x8 al[TM], bl[TN];
      const unsigned char* Al = As + A_BYTES + B_BYTES;
      const unsigned char* Bl = Bs + A_BYTES + B_BYTES;
#pragma unroll
      for (int i = 0; i < TM; ++i) al[i] = *reinterpret_cast<const bf16x8*>(Al + i * 16 * LDS_ROW);
#pragma unroll
      for (int j = 0; j < TN; ++j) bl[j] = *reinterpret_cast<const bf16x8*>(Bl + j * 16 * LDS_ROW);
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
        if (g.ep.bias && lead) v += g.ep.bias[n];
        if (g.ep.act == 1) v = fmaxf(v, 0.f);
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

template <int BM, int BN, bool PRECISE>
void launch_layouts(const GemmArgs& g, dim3 grid, hipStream_t st) {
  if (g.A.kmajor && g.B.kmajor)
    hipLaunchKernelGGL((gemm_kernel<BM, BN, true, true, PRECISE>), grid, dim3(256), 0, st, g);
  else if (g.A.kmajor && !g.B.kmajor)
    hipLaunchKernelGGL((gemm_kernel<BM, BN, true, false, PRECISE>), grid, dim3(256), 0, st, g);
  else if (!g.A.kmajor && g.B.kmajor)
    hipLaunchKernelGGL((gemm_kernel<BM, BN, false, true, PRECISE>), grid, dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL((gemm_kernel<BM, BN, false, false, PRECISE>), grid, dim3(256), 0, st, g);
}

bool vec_ok(const GemmOperand& o) {
  return ((uintptr_t)o.p % 16 == 0) && (o.sp.ld % 4 == 0) && (o.sp.bs % 4 == 0) &&
         (o.zo % 4 == 0) && (o.zi % 4 == 0);
}

}  // namespace

// ---- optional per-launch timing (bench.py roofline leg): HIP events on the launch stream ----
namespace {
struct ProfRec { hipEvent_t a, b; double flops; };
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
}  // namespace

void s2st_gemm_profile_enable(int on) { g_prof_on = on != 0; }

int s2st_gemm_profile_read(double* flops, double* ms, long* launches) {
  double f = 0, t = 0;
  for (auto& r : g_prof) {
    hipEventSynchronize(r.b);
    float e = 0.f;
    hipEventElapsedTime(&e, r.a, r.b);
    t += e;
    f += r.flops;
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  *flops = f;
  *ms = t;
  *launches = (long)g_prof.size();
  g_prof.clear();
  return 0;
}

int s2st_gemm(GemmArgs g, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return 0;
  ProfRec rec{};
  if (g_prof_on) {
    hipEventCreate(&rec.a);
    hipEventCreate(&rec.b);
    rec.flops = 2.0 * g.M * g.N * (double)g.K * g.batch;
    hipEventRecord(rec.a, st);
  }
  if (g.zdiv <= 0) g.zdiv = 1;
  g.avec = vec_ok(g.A) ? 1 : 0;
  g.bvec = vec_ok(g.B) ? 1 : 0;
  auto tiles = [&](int bm, int bn) {
    return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * g.batch;
  };
  bool big = !g.precise && tiles(128, 128) >= 192;
  int bm = big ? 128 : 64, bn = bm;
  long nt = tiles(bm, bn);
  // split-K only for accumulating outputs (weight gradients): partial sums are added with
  // fp32 atomics, the epilogue must then be linear.
  g.splitk = 1;
  bool linear_epi = !g.ep.act && g.ep.drop_p == 0.f;
  if (g.ep.accumulate && linear_epi && nt < 256 && g.K >= 8 * BK) {
    int want = (int)((512 + nt - 1) / nt);
    int maxs = g.K / (4 * BK);
    g.splitk = want < maxs ? want : maxs;
    if (g.splitk < 1) g.splitk = 1;
  }
  int kt = (g.K + BK - 1) / BK;
  g.kchunk = ((kt + g.splitk - 1) / g.splitk) * BK;
  g.splitk = (g.K + g.kchunk - 1) / g.kchunk;
  if (g.splitk < 1) g.splitk = 1;
  dim3 grid((g.N + bn - 1) / bn, (g.M + bm - 1) / bm, g.batch * g.splitk);
  if (grid.y > 65535 || grid.z > 65535) return -2;
  if (g.precise) {
    launch_layouts<64, 64, true>(g, grid, st);
  } else if (big) {
    launch_layouts<128, 128, false>(g, grid, st);
  } else {
    launch_layouts<64, 64, false>(g, grid, st);
  }
  if (g_prof_on) {
    hipEventRecord(rec.b, st);
    g_prof.push_back(rec);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
