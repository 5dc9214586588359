// Element-wise / per-channel kernels of the s2st hot path (HBM-bound): GLU, positional
// embedding add, embedding gather/scatter, dropout, BatchNorm1d (batch statistics over all
// B*T rows, tanh, dropout), conv-weight re-layout, row copies into halo-padded buffers.
//
// Reference call sites replaced: s2st_transformer.py:135-139 (F.glu), :197-208, :385-387
// (scale + sinusoidal positions + dropout), fairseq/models/text_to_speech/tacotron2.py:101-126
// (BatchNorm1d + tanh + dropout in the post-net), transformer_decoder.py:303-320 (token
// embedding), fairseq/modules/fairseq_dropout.py:16-27.
#include <cstdlib>

#include "s2st_ops.h"
#include "s2st_prof.h"

namespace {

constexpr int EW_BLOCK = 256;

inline unsigned ew_grid(long n, int per_thread = 1) {
  long b = (n + (long)EW_BLOCK * per_thread - 1) / ((long)EW_BLOCK * per_thread);
  return (unsigned)(b < 1 ? 1 : b);
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ x, Split xsp,
                                                        float* __restrict__ y, Split ysp, int rows,
                                                        int C4) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  long n = (long)rows * C4;
  if (i >= n) return;
  int r = (int)(i / C4), c = (int)(i - (long)r * C4);
  reinterpret_cast<float4*>(y + split_off(ysp, r))[c] =
      reinterpret_cast<const float4*>(x + split_off(xsp, r))[c];
}

// bf16 rows (4 elements = 8 bytes per thread)
__global__ __launch_bounds__(256) void copy_rows_bf16_kernel(const uint16_t* __restrict__ x, Split xsp,
                                                             uint16_t* __restrict__ y, Split ysp, int rows,
                                                             int C4) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  long n = (long)rows * C4;
  if (i >= n) return;
  int r = (int)(i / C4), c = (int)(i - (long)r * C4);
  reinterpret_cast<uint2*>(y + split_off(ysp, r))[c] = reinterpret_cast<const uint2*>(x + split_off(xsp, r))[c];
}

// The conv backward's operand image in ONE pass: y [B][Th][O] bf16 = x's rows at u = pad + stride * t, zeros everywhere
// else (halos and, for stride 2, the stuffed rows) -- every 16-byte chunk of y is written exactly once (was: a memset of
// the image + a row copy).  O % 8 == 0.
__global__ __launch_bounds__(256) void halo_image_bf16_kernel(const uint16_t* __restrict__ x, long ldx,
                                                              uint16_t* __restrict__ y, int B, int Tout, int Th, int O8,
                                                              int pad, int stride) {
  const long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  if (i >= (long)B * Th * O8) return;
  const int c = (int)(i % O8);
  const long bu = i / O8;
  const int u = (int)(bu % Th), b = (int)(bu / Th);
  const int d = u - pad;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (d >= 0 && d % stride == 0 && d / stride < Tout)
    v = reinterpret_cast<const uint4*>(x + ((long)b * Tout + d / stride) * ldx)[c];
  reinterpret_cast<uint4*>(y)[i] = v;
}

// bf16 twin of a halo image [B][T + 2 pad][C]: interior rows converted from the fp32 image, halo rows written as zeros
// WITHOUT reading the fp32 halos (fast mode never reads those, so the fp32 image is not cleared any more: one pass
// instead of memset + cast).  C % 4 == 0.
// (plain != 0: x is the plain rows [B * T][C] instead of the fp32 image -- the first convolution of a stack)
__global__ __launch_bounds__(256) void cast_bf16_halo_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int B,
                                                             int T, int pad, int C4, int plain) {
  const long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  const int Th = T + 2 * pad;
  if (i >= (long)B * Th * C4) return;
  const long bu = i / C4;
  const int u = (int)(bu % Th);
  uint2 v = make_uint2(0u, 0u);
  if (u >= pad && u < pad + T) {
    const long src = plain ? ((bu / Th) * T + (u - pad)) * C4 + (i - bu * C4) : i;
    const float4 f = reinterpret_cast<const float4*>(x)[src];
    v = pack_bf16x4(f.x, f.y, f.z, f.w);
  }
  reinterpret_cast<uint2*>(y)[i] = v;
}

// fp32 [rows][cols] (row stride ldx) -> bf16 [rows][ldy], columns [cols, ldy) zero-filled.
// One thread per 4 output columns (ldy % 4 == 0); 16-byte loads when the source row allows.
__global__ __launch_bounds__(256) void cast_bf16_rows_kernel(const float* __restrict__ x, long ldx,
                                                             uint16_t* __restrict__ y, long ldy, long rows,
                                                             int cols, int vec) {
  const long q = ldy >> 2;
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  if (i >= rows * q) return;
  const long r = i / q;
  const int c = (int)(i - r * q) * 4;
  const float* xr = x + r * ldx;
  float v[4];
  if (vec && c + 3 < cols) {
    float4 f = *reinterpret_cast<const float4*>(xr + c);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = c + e < cols ? xr[c + e] : 0.f;
  }
  *reinterpret_cast<uint2*>(y + r * ldy + c) = pack_bf16x4(v[0], v[1], v[2], v[3]);
}

// bf16 [R][C] -> [C][R] through a 64 x 64 LDS tile (pre-transposed weight copies for the data-gradient GEMMs)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                             int R, int C) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? x[(long)(r0 + r) * C + c0 + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < R && c0 + c < C) y[(long)(c0 + c) * R + r0 + r] = tile[r][c];
  }
}

// All 2-D weights of the arena in ONE launch (the per-matrix form above cost ~108 launches of ~10 us per step): the
// table of matrices rides in the kernel arguments; a workgroup finds its matrix by its tile index.  Rows and columns
// are multiples of 8 (reg_wt), so both sides move 16-byte chunks.
__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const uint16_t* __restrict__ xb,
                                                                     uint16_t* __restrict__ yb, s2st_transpose_table t) {
  __shared__ uint16_t tile[64][72];
  int lo = 0, hi = t.n - 1;
  const unsigned id = blockIdx.x;
  while (lo < hi) {  // last entry with tile0 <= id
    const int mid = (lo + hi + 1) >> 1;
    if (t.tile0[mid] <= id) lo = mid; else hi = mid - 1;
  }
  const int R = t.rows8[lo] * 8, C = t.cols8[lo] * 8;
  const uint16_t* x = xb + t.off[lo];
  uint16_t* y = yb + t.off[lo];
  const int local = (int)(id - t.tile0[lo]), tc = (C + 63) / 64;
  const int r0 = (local / tc) * 64, c0 = (local % tc) * 64;
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {
    const int r = i >> 3, c = (i & 7) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < R && c0 + c < C) v = *reinterpret_cast<const uint4*>(x + (long)(r0 + r) * C + c0 + c);
    *reinterpret_cast<uint4*>(&tile[r][c]) = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {
    const int c = i >> 3, r = (i & 7) * 8;
    if (r0 + r < R && c0 + c < C) {
      unsigned w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (unsigned)tile[r + 2 * j][c] | ((unsigned)tile[r + 2 * j + 1][c] << 16);
      *reinterpret_cast<uint4*>(y + (long)(c0 + c) * R + r0 + r) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__global__ __launch_bounds__(256) void glu_fwd_kernel(const float* __restrict__ a,
                                                      float* __restrict__ y, Split ysp, int rows,
                                                      int C) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  long n = (long)rows * C;
  if (i >= n) return;
  int r = (int)(i / C), c = (int)(i - (long)r * C);
  const float* ar = a + (long)r * 2 * C;
  y[split_off(ysp, r) + c] = ar[c] * sigmoidf_(ar[c + C]);
}

__global__ __launch_bounds__(256) void glu_bwd_kernel(const float* __restrict__ a,
                                                      const float* __restrict__ dy, Split dysp,
                                                      float* __restrict__ da, Split dasp, int rows,
                                                      int C, uint16_t* __restrict__ dah, long ldh) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  long n = (long)rows * C;
  if (i >= n) return;
  int r = (int)(i / C), c = (int)(i - (long)r * C);
  const float* ar = a + (long)r * 2 * C;
  float g = dy[split_off(dysp, r) + c];
  float s = sigmoidf_(ar[c + C]);
  float* dr = da + split_off(dasp, r);
  const float d0 = g * s, d1 = g * ar[c] * s * (1.f - s);
  dr[c] = d0;
  dr[c + C] = d1;
  if (dah) {  // bf16 twin (the conv backward's GEMM operand)
    const unsigned pk = pack_bf16x4(d0, d1, 0.f, 0.f).x;
    dah[(long)r * ldh + c] = (uint16_t)(pk & 0xffffu);
    dah[(long)r * ldh + c + C] = (uint16_t)(pk >> 16);
  }
}

// spk_table / spk_ids (optional): the utterance's speaker embedding row is added to EVERY position of the utterance,
// padded ones included, before the dropout (s2st_transformer.py:203-208); T = rows per utterance
__global__ __launch_bounds__(256) void add_pe_kernel(const float* __restrict__ x,
                                                     float* __restrict__ y,
                                                     const int* __restrict__ pos,
                                                     const float* __restrict__ table, int rows,
                                                     int C, float scale,
                                                     const float* __restrict__ alpha_ptr,
                                                     float drop_p, uint64_t seed,
                                                     const float* __restrict__ spk_table,
                                                     const long* __restrict__ spk_ids, int T) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  long n = (long)rows * C;
  if (i >= n) return;
  int r = (int)(i / C), c = (int)(i - (long)r * C);
  float alpha = alpha_ptr ? alpha_ptr[0] : 1.f;
  float v = scale * x[i] + alpha * table[(long)pos[r] * C + c];
  if (spk_table) v += spk_table[spk_ids[r / T] * C + c];
  if (drop_p > 0.f) v *= drop_scale(seed, (uint64_t)i, drop_p, 1.f / (1.f - drop_p));
  y[i] = v;
}

// Gradient of a speaker-embedding table whose row ids[b] was added to (T_sum > 1: every one of the first T_sum rows
// of) utterance b's block of `T` rows: dtable[s][c] += sum over {b : ids[b] == s}, t < T_sum of mask(b, t, c) * dy[b][t][c].
// One thread per (speaker, column), utterances and rows in index order: no atomics, run-to-run identical.
__global__ __launch_bounds__(256) void speaker_bwd_kernel(const float* __restrict__ dy, const long* __restrict__ ids, int B,
                                                          int T, int T_sum, int C, int n_spk, float drop_p, uint64_t seed,
                                                          float* __restrict__ dtable) {
  const int c = blockIdx.x * 256 + threadIdx.x, sp = blockIdx.y;
  if (c >= C || sp >= n_spk) return;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float a = 0.f;
  for (int b = 0; b < B; ++b) {
    if (ids[b] != sp) continue;
    for (int t = 0; t < T_sum; ++t) {
      const long i = ((long)b * T + t) * C + c;
      float g = dy[i];
      if (drop_p > 0.f) g *= drop_scale(seed, (uint64_t)i, drop_p, inv_keep);
      a += g;
    }
  }
  dtable[(long)sp * C + c] += a;
}

// y[b][t0][:] = table[ids[b]][:] for every utterance (the decoder's first input frame, s2st_transformer.py:441-444)
__global__ __launch_bounds__(256) void speaker_set_rows_kernel(const float* __restrict__ table, const long* __restrict__ ids,
                                                               float* __restrict__ y, int B, int T, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * C) return;
  const int b = (int)(i / C), c = (int)(i - (long)b * C);
  y[(long)b * T * C + c] = table[ids[b] * C + c];
}

// t2s text encoder (t2s_transformer.py:107-111): columns [coff, coff + Sd) of every row of utterance b <- table[ids[b]]
__global__ __launch_bounds__(256) void speaker_fill_cols_kernel(const float* __restrict__ table, const long* __restrict__ ids,
                                                                float* __restrict__ y, int B, int T, int ld, int coff, int Sd) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * T * Sd) return;
  const long r = i / Sd;
  const int c = (int)(i - r * Sd);
  y[r * ld + coff + c] = table[ids[r / T] * Sd + c];
}

// ... and its gradient: dtable[s][c] += sum over {b : ids[b] == s}, t < T of dy[b][t][coff + c] -- one thread per
// (speaker, column), utterances and rows in index order (no atomics)
__global__ __launch_bounds__(256) void speaker_cols_bwd_kernel(const float* __restrict__ dy, const long* __restrict__ ids, int B,
                                                               int T, int ld, int coff, int Sd, int n_spk,
                                                               float* __restrict__ dtable) {
  const int c = blockIdx.x * 256 + threadIdx.x, sp = blockIdx.y;
  if (c >= Sd || sp >= n_spk) return;
  float a = 0.f;
  for (int b = 0; b < B; ++b) {
    if (ids[b] != sp) continue;
    for (int t = 0; t < T; ++t) a += dy[((long)b * T + t) * ld + coff + c];
  }
  dtable[(long)sp * Sd + c] += a;
}

// dx[r][0..C) (+)= g[r][0..C) for g rows of stride ldg
__global__ __launch_bounds__(256) void split_cols_kernel(const float* __restrict__ g, int ldg, float* __restrict__ dx, int ldx,
                                                         int rows, int C, int acc) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * C) return;
  const long r = i / C;
  const int c = (int)(i - r * C);
  const float v = g[r * ldg + c];
  float* d = dx + r * ldx + c;
  *d = acc ? *d + v : v;
}

// dalpha += sum_i dy[i] * mask(i) * PE[pos(row)][c]
__global__ __launch_bounds__(256) void pe_alpha_bwd_kernel(const float* __restrict__ dy,
                                                           const int* __restrict__ pos,
                                                           const float* __restrict__ table,
                                                           int rows, int C, float drop_p,
                                                           uint64_t seed,
                                                           float* __restrict__ dalpha, float* __restrict__ part) {
  __shared__ float red[4];
  long n = (long)rows * C;
  float a = 0.f;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * EW_BLOCK) {
    int r = (int)(i / C), c = (int)(i - (long)r * C);
    float g = dy[i];
    if (drop_p > 0.f) g *= drop_scale(seed, (uint64_t)i, drop_p, inv_keep);
    a += g * table[(long)pos[r] * C + c];
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = red[0] + red[1] + red[2] + red[3];
    if (part) part[blockIdx.x] = v;  // folded in block order by the caller (no atomics)
    else atomicAdd(dalpha, v);
  }
}

__global__ __launch_bounds__(256) void embed_fwd_kernel(const long* __restrict__ tok,
                                                        const float* __restrict__ table,
                                                        float* __restrict__ y, int rows, int C,
                                                        float scale) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  long n = (long)rows * C;
  if (i >= n) return;
  int r = (int)(i / C), c = (int)(i - (long)r * C);
  y[i] = scale * table[tok[r] * C + c];
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const long* __restrict__ tok,
                                                        const float* __restrict__ dy,
                                                        float* __restrict__ dtable, int rows, int C,
                                                        float scale, long pad) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  long n = (long)rows * C;
  if (i >= n) return;
  int r = (int)(i / C), c = (int)(i - (long)r * C);
  long t = tok[r];
  if (t == pad) return;  // nn.Embedding(padding_idx): no gradient for the pad row
  atomicAdd(&dtable[t * C + c], scale * dy[i]);
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x,
                                                      float* __restrict__ y, long n, float a,
                                                      float p, uint64_t seed, int accumulate) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  if (i >= n) return;
  float v = a * x[i];
  if (p > 0.f) v *= drop_scale(seed, (uint64_t)i, p, 1.f / (1.f - p));
  y[i] = accumulate ? y[i] + v : v;
}

__global__ __launch_bounds__(256) void relu_drop_bwd_kernel(const float* __restrict__ dy,
                                                            const float* __restrict__ yout,
                                                            float* __restrict__ dz, long n,
                                                            float inv_keep) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  if (i >= n) return;
  dz[i] = yout[i] != 0.f ? dy[i] * inv_keep : 0.f;
}

// Backward prologue of y = dropout(act(x W^T + b)): dpre = f(dy) written as the bf16 GEMM operand
// (row stride ldp, pad columns zeroed) and, fused, the bias gradient db += colsum(dpre).
//   MODE 0: dpre = dy ; 1: ReLU + dropout from the OUTPUT y (y != 0 ? dy/(1-p) : 0) ;
//   2: dropout mask regenerated from (seed, element index)
// 256 threads = 4 waves; a wave owns every 4th row of the block's row slab, a lane 4 columns.
template <int MODE>
__global__ __launch_bounds__(256) void dpre_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                   const uint16_t* __restrict__ yb, uint16_t* __restrict__ dph, long ldp, float* __restrict__ dpre,
                                                   float* __restrict__ dbias, int M, int N, int rows_per_block,
                                                   float p, float inv_keep, uint64_t seed, float* __restrict__ part) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < N) {
    // 4 rows per pass with all loads issued first (the kernel is latency-bound otherwise)
    for (int rb = r0 + wave; rb < r1; rb += 16) {
      float4 d[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = rb + 4 * u;
        d[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        yv[u] = d[u];
        if (r < r1) {
          d[u] = *reinterpret_cast<const float4*>(dy + (long)r * N + c);
          if (MODE == 1) {
            if (y) yv[u] = *reinterpret_cast<const float4*>(y + (long)r * N + c);
            else {  // only the bf16 copy of the layer output exists: +-0 <=> fp32 zero (bf16 keeps the exponent range)
              const uint2 q = *reinterpret_cast<const uint2*>(yb + (long)r * N + c);
              yv[u] = make_float4((float)(q.x & 0x7fffu), (float)((q.x >> 16) & 0x7fffu), (float)(q.y & 0x7fffu),
                                  (float)((q.y >> 16) & 0x7fffu));
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = rb + 4 * u;
        if (r >= r1) continue;
        const long o = (long)r * N + c;
        float4 dd = d[u];
        if (MODE == 1) {
          dd.x = yv[u].x != 0.f ? dd.x * inv_keep : 0.f; dd.y = yv[u].y != 0.f ? dd.y * inv_keep : 0.f;
          dd.z = yv[u].z != 0.f ? dd.z * inv_keep : 0.f; dd.w = yv[u].w != 0.f ? dd.w * inv_keep : 0.f;
        } else if (MODE == 2) {
          dd.x *= drop_scale(seed, (uint64_t)o, p, inv_keep); dd.y *= drop_scale(seed, (uint64_t)o + 1, p, inv_keep);
          dd.z *= drop_scale(seed, (uint64_t)o + 2, p, inv_keep); dd.w *= drop_scale(seed, (uint64_t)o + 3, p, inv_keep);
        }
        *reinterpret_cast<uint2*>(dph + (long)r * ldp + c) = pack_bf16x4(dd.x, dd.y, dd.z, dd.w);
        if (dpre) *reinterpret_cast<float4*>(dpre + o) = dd;
        a0 += dd.x; a1 += dd.y; a2 += dd.z; a3 += dd.w;
      }
    }
  } else if (c < ldp) {
    for (int r = r0 + wave; r < r1; r += 4) *reinterpret_cast<uint2*>(dph + (long)r * ldp + c) = make_uint2(0, 0);
  }
  if (!dbias) return;
  red[wave][lane * 4 + 0] = a0; red[wave][lane * 4 + 1] = a1;
  red[wave][lane * 4 + 2] = a2; red[wave][lane * 4 + 3] = a3;
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < N) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (part) part[(long)blockIdx.y * N + cc] = v;  // folded in slab order by the launcher's second kernel
    else atomicAdd(dbias + cc, v);
  }
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x,
                                                   float* __restrict__ y, long n, float a) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n, float a) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  if (i < n) x[i] *= a;
}

// W[O][I][Kw] -> Wf[O][Kw][I] ; Wd[I][Kw-1-j][O]
__global__ __launch_bounds__(256) void conv_w_permute_kernel(const float* __restrict__ w,
                                                             float* __restrict__ wf,
                                                             float* __restrict__ wd, int O, int I,
                                                             int Kw, uint16_t* __restrict__ wfh,
                                                             uint16_t* __restrict__ wdh) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  long n = (long)O * I * Kw;
  if (i >= n) return;
  int j = (int)(i % Kw);
  int c = (int)((i / Kw) % I);
  int o = (int)(i / ((long)Kw * I));
  float v = w[i];
  const uint16_t vh = (uint16_t)(pack_bf16x4(v, 0.f, 0.f, 0.f).x & 0xffffu);  // bf16 twins of the same layouts
  if (wf) wf[((long)o * Kw + j) * I + c] = v;
  if (wd) wd[((long)c * Kw + (Kw - 1 - j)) * O + o] = v;
  if (wfh) wfh[((long)o * Kw + j) * I + c] = vh;
  if (wdh) wdh[((long)c * Kw + (Kw - 1 - j)) * O + o] = vh;
}

// The same re-layout through a 32 x 32 LDS tile per (o block, c block, tap j): reads walk c (stride Kw floats), the Wf
// stores walk c, the Wd stores walk o -- every global access of a wave is one or a few contiguous runs (the per-element
// kernel above scatters 2- and 4-byte stores at strides of I and O elements: 35 us for the 512 x 512 x 5 post-net
// weights, seven times a step)
__global__ __launch_bounds__(256) void conv_w_permute_tiled_kernel(const float* __restrict__ w, float* __restrict__ wf,
                                                                   float* __restrict__ wd, int O, int I, int Kw,
                                                                   uint16_t* __restrict__ wfh, uint16_t* __restrict__ wdh) {
  __shared__ float tile[32][33];
  const int j = blockIdx.z, o0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int o = o0 + r, c = c0 + tx;
    float v = 0.f;
    if (o < O && c < I) {
      v = w[((long)o * I + c) * Kw + j];
      const long i = ((long)o * Kw + j) * I + c;
      if (wf) wf[i] = v;
      if (wfh) wfh[i] = (uint16_t)(pack_bf16x4(v, 0.f, 0.f, 0.f).x & 0xffffu);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  if (!wd && !wdh) return;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, o = o0 + tx;
    if (o < O && c < I) {
      const float v = tile[tx][r];
      const long i = ((long)c * Kw + (Kw - 1 - j)) * O + o;
      if (wd) wd[i] = v;
      if (wdh) wdh[i] = (uint16_t)(pack_bf16x4(v, 0.f, 0.f, 0.f).x & 0xffffu);
    }
  }
}

// (slabs > 1: dwf is [slabs][O][Kw][I] -- partial sums over K ranges of the token axis, added in slab order)
__global__ __launch_bounds__(256) void conv_w_unpermute_acc_kernel(const float* __restrict__ dwf,
                                                                   float* __restrict__ dw, int O,
                                                                   int I, int Kw, int slabs) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  long n = (long)O * I * Kw;
  if (i >= n) return;
  int j = (int)(i % Kw);
  int c = (int)((i / Kw) % I);
  int o = (int)(i / ((long)Kw * I));
  const long src = ((long)o * Kw + j) * I + c;
  float v = dwf[src];
  for (int sl = 1; sl < slabs; ++sl) v += dwf[sl * n + src];
  dw[i] += v;
}

// ---- column reductions with a per-element functor returning two values ------------------
// Deterministic: row slabs write partial sums to scratch and a second small kernel folds them in slab order
// (BatchNorm statistics feed every later activation, and with fp32 atomics their last-bit noise is amplified to
// ~1e-3 by the bf16 operand rounding downstream; a ticket / last-block fold in one launch needs device-scope
// fences, which on this multi-L2 chip cost more than the extra launch).  scratch: part [slabs][2][cols] floats.
constexpr int CR_MAX_SLABS = 64;
template <class F>
__global__ __launch_bounds__(256) void colreduce2_kernel(F f, int rows, int cols, int rows_per_block,
                                                         float* __restrict__ part) {
  __shared__ float red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  float a0 = 0.f, a1 = 0.f;
  if (c < cols)
    for (int r = r0 + ty; r < r1; r += 4) {
      float2 v = f(r, c);
      a0 += v.x;
      a1 += v.y;
    }
  red[0][ty][tx] = a0;
  red[1][ty][tx] = a1;
  __syncthreads();
  if (ty == 0 && c < cols) {
    part[((long)blockIdx.y * 2 + 0) * cols + c] = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
    part[((long)blockIdx.y * 2 + 1) * cols + c] = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
  }
}
// out = fold of the slab partials; optionally also acc += out (parameter gradients)
__global__ __launch_bounds__(256) void colreduce2_fold_kernel(const float* __restrict__ part, int slabs, int cols,
                                                              float* __restrict__ out0, float* __restrict__ out1,
                                                              float* __restrict__ acc0, float* __restrict__ acc1) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  // loads in batches of 16 slabs issued together (a dependent load-add chain would cost one memory latency
  // per slab); the adds stay in slab order
  float s0 = 0.f, s1 = 0.f;
  for (int sb = 0; sb < slabs; sb += 16) {
    float v0[16], v1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int s = min(sb + j, slabs - 1);
      v0[j] = part[((long)s * 2 + 0) * cols + c];
      v1[j] = part[((long)s * 2 + 1) * cols + c];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (sb + j < slabs) {
        s0 += v0[j];
        s1 += v1[j];
      }
  }
  out0[c] = s0;
  if (out1) out1[c] = s1;
  if (acc0) acc0[c] += s0;
  if (acc1) acc1[c] += s1;
}

// scratch = tmp + 2 * cols (s2st_bn_* pass tmp of S2ST_BN_TMP_FLOATS(C) floats)
// out0 == nullptr: leave the fold to the caller's next kernel (*slabs_out partials per column)
template <class F>
int colreduce2(F f, int rows, int cols, float* out0, float* out1, float* scratch, hipStream_t st,
               float* acc0 = nullptr, float* acc1 = nullptr, int* slabs_out = nullptr) {
  int cb = (cols + 63) / 64;
  int slabs = (512 + cb - 1) / cb;
  if (slabs > CR_MAX_SLABS) slabs = CR_MAX_SLABS;
  int rpb = (rows + slabs - 1) / slabs;
  if (rpb < 16) rpb = 16;
  slabs = (rows + rpb - 1) / rpb;
  S2ST_LAUNCH((colreduce2_kernel<F>), dim3(cb, slabs), dim3(256), 0, st, f, rows, cols, rpb, scratch);
  if (slabs_out) *slabs_out = slabs;
  if (out0)
    S2ST_LAUNCH(colreduce2_fold_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, (const float*)scratch, slabs,
                       cols, out0, out1, acc0, acc1);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

// One pass over x: d = x - x[row 0] (a data sample of the column: |mean - shift| is a few standard deviations at most, so
// var = E[d^2] - E[d]^2 loses ~10 ulp where the unshifted form can lose everything), returns {d, d^2}
struct ShiftSqF {
  const float* x; int C;
  __device__ float2 operator()(int r, int c) const {
    const float d = x[(long)r * C + c] - x[c];
    return make_float2(d, d * d);
  }
};
// part: the slab partials [slabs][2][C] of {d, d^2} (folded here, in slab order); x0 = row 0 of x (the shift)
__global__ __launch_bounds__(256) void bn_finalize_shift_kernel(const float* __restrict__ x0, const float* __restrict__ part,
                                                                int slabs, float* __restrict__ mean,
                                                                float* __restrict__ var, float* __restrict__ run_mean,
                                                                float* __restrict__ run_var, int C, int rows,
                                                                float momentum) {
  int c = blockIdx.x * EW_BLOCK + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int sb = 0; sb < slabs; sb += 16) {
    float v1[16], v2[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const long s = min(sb + j, slabs - 1);
      v1[j] = part[(s * 2 + 0) * C + c];
      v2[j] = part[(s * 2 + 1) * C + c];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (sb + j < slabs) { s1 += v1[j]; s2 += v2[j]; }
  }
  const float d1 = s1 / rows;
  const float m = x0[c] + d1;
  const float v = fmaxf(s2 / rows - d1 * d1, 0.f);
  mean[c] = m;
  var[c] = v;
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * m;
    float unb = rows > 1 ? v * ((float)rows / (float)(rows - 1)) : v;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * unb;
  }
}


__global__ __launch_bounds__(256) void bn_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ var,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
    Split ysp, const float* __restrict__ resid, int rows, int C, float eps, int tanh_, float drop_p,
    uint64_t seed) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  long n = (long)rows * C;
  if (i >= n) return;
  int r = (int)(i / C), c = (int)(i - (long)r * C);
  float u = gamma[c] * (x[i] - mean[c]) * rsqrtf(var[c] + eps) + beta[c];
  if (tanh_ == 1) u = tanhf(u);
  else if (tanh_ == 2) u = fmaxf(u, 0.f);  // ReLU (t2s encoder prenet)
  if (drop_p > 0.f) u *= drop_scale(seed, (uint64_t)i, drop_p, 1.f / (1.f - drop_p));
  if (resid) u += resid[i];
  y[split_off(ysp, r) + c] = u;
}

// The same transform written as the NEXT convolution's operand: a bf16 halo image [B][T + 2 pad][C] (zero halos, one
// 8-byte store per 4 channels) and, optionally, the fp32 result rows y [B * T][C] -- no fp32 image, no memset, no cast pass
__global__ __launch_bounds__(256) void bn_apply_img_kernel(
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ var,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, uint16_t* __restrict__ img,
    int B, int T, int pad, int C4, float eps, int tanh_, float drop_p, uint64_t seed) {
  const long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  const int Th = T + 2 * pad;
  if (i >= (long)B * Th * C4) return;
  const int c = (int)(i % C4) * 4;
  const long bu = i / C4;
  const int u = (int)(bu % Th), b = (int)(bu / Th);
  uint2 h = make_uint2(0u, 0u);
  if (u >= pad && u < pad + T) {
    const long e0 = ((long)b * T + (u - pad)) * (4L * C4) + c;  // element index in the plain [B * T][C] rows
    const float4 xv = *reinterpret_cast<const float4*>(x + e0);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = gamma[c + k] * (xs[k] - mean[c + k]) * rsqrtf(var[c + k] + eps) + beta[c + k];
      if (tanh_ == 1) t = tanhf(t);
      else if (tanh_ == 2) t = fmaxf(t, 0.f);
      if (drop_p > 0.f) t *= drop_scale(seed, (uint64_t)(e0 + k), drop_p, 1.f / (1.f - drop_p));
      o[k] = t;
    }
    if (y) *reinterpret_cast<float4*>(y + e0) = make_float4(o[0], o[1], o[2], o[3]);
    h = pack_bf16x4(o[0], o[1], o[2], o[3]);
  }
  reinterpret_cast<uint2*>(img)[i] = h;
}

// du = dy * dropmask * (1 - tanh^2)   (recomputed from x);  returns {du, du * xhat}
struct BnBwdF {
  const float* dy; Split dysp; const float* x; const float* mean; const float* var;
  const float* gamma; const float* beta; int C; float eps; int tanh_; float drop_p; uint64_t seed;
  __device__ float2 operator()(int r, int c) const {
    long i = (long)r * C + c;
    float xh = (x[i] - mean[c]) * rsqrtf(var[c] + eps);
    float g = dy[split_off(dysp, r) + c];
    if (drop_p > 0.f) g *= drop_scale(seed, (uint64_t)i, drop_p, 1.f / (1.f - drop_p));
    if (tanh_ == 1) {
      float t = tanhf(gamma[c] * xh + beta[c]);
      g *= (1.f - t * t);
    } else if (tanh_ == 2) {
      g = (gamma[c] * xh + beta[c]) > 0.f ? g : 0.f;
    }
    return make_float2(g, g * xh);
  }
};

__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(BnBwdF f, const float* __restrict__ sums,
                                                        float* __restrict__ dx, Split dxsp,
                                                        int rows, int C, uint16_t* __restrict__ dxh, long ldh) {
  long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x;
  long n = (long)rows * C;
  if (i >= n) return;
  int r = (int)(i / C), c = (int)(i - (long)r * C);
  float2 v = f(r, c);  // {du, du*xhat}
  float rstd = rsqrtf(f.var[c] + f.eps);
  float xh = (f.x[i] - f.mean[c]) * rstd;
  float inv = 1.f / rows;
  const float d = f.gamma[c] * rstd * (v.x - sums[c] * inv - xh * sums[C + c] * inv);
  dx[split_off(dxsp, r) + c] = d;
  if (dxh) dxh[(long)r * ldh + c] = (uint16_t)(pack_bf16x4(d, 0.f, 0.f, 0.f).x & 0xffffu);  // bf16 twin
}

__global__ __launch_bounds__(256) void add_vec_kernel(const float* __restrict__ a,
                                                      float* __restrict__ b, int n) {
  int i = blockIdx.x * EW_BLOCK + threadIdx.x;
  if (i < n) b[i] += a[i];
}

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH)

int s2st_copy_rows(const float* x, Split xsp, float* y, Split ysp, int rows, int C,
                   hipStream_t st) {
  if (rows <= 0) return 0;
  if (C % 4) return S2ST_ERR_SHAPE;
  long n = (long)rows * (C / 4);
  S2ST_LAUNCH(copy_rows_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, xsp, y, ysp, rows, C / 4);
  return LAUNCH_OK();
}

int s2st_copy_rows_bf16(const uint16_t* x, Split xsp, uint16_t* y, Split ysp, int rows, int C, hipStream_t st) {
  if (rows <= 0 || C <= 0) return 0;
  if (C % 4 || xsp.ld % 4 || ysp.ld % 4 || xsp.bs % 4 || ysp.bs % 4 || ((uintptr_t)x % 8) || ((uintptr_t)y % 8))
    return S2ST_ERR_SHAPE;
  long n = (long)rows * (C / 4);
  S2ST_LAUNCH(copy_rows_bf16_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, xsp, y, ysp, rows, C / 4);
  return LAUNCH_OK();
}

int s2st_halo_image_bf16(const uint16_t* x, long ldx, uint16_t* y, int B, int Tout, int Th, int O, int pad, int stride,
                         hipStream_t st) {
  if (B <= 0 || Th <= 0 || O <= 0) return 0;
  if (O % 8 || ldx % 8 || stride < 1 || ((uintptr_t)x % 16) || ((uintptr_t)y % 16)) return S2ST_ERR_SHAPE;
  const long n = (long)B * Th * (O / 8);
  S2ST_LAUNCH(halo_image_bf16_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, ldx, y, B, Tout, Th, O / 8, pad, stride);
  return LAUNCH_OK();
}

int s2st_cast_bf16_halo(const float* x, uint16_t* y, int B, int T, int pad, int C, hipStream_t st, int plain) {
  if (B <= 0 || T + 2 * pad <= 0 || C <= 0) return 0;
  if (C % 4 || ((uintptr_t)x % 16) || ((uintptr_t)y % 8)) return S2ST_ERR_SHAPE;
  const long n = (long)B * (T + 2 * pad) * (C / 4);
  S2ST_LAUNCH(cast_bf16_halo_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, y, B, T, pad, C / 4, plain);
  return LAUNCH_OK();
}

int s2st_cast_bf16_rows(const float* x, long ldx, uint16_t* y, long ldy, long rows, int cols, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  if (ldy % 4 != 0 || ldy < cols) return S2ST_ERR_SHAPE;
  const int vec = ((uintptr_t)x % 16 == 0) && (ldx % 4 == 0);
  long n = rows * (ldy >> 2);
  S2ST_LAUNCH(cast_bf16_rows_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, ldx, y, ldy, rows, cols, vec);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_transpose_bf16(const uint16_t* x, uint16_t* y, int R, int C, hipStream_t st) {
  if (R <= 0 || C <= 0) return 0;
  S2ST_LAUNCH(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, st, x, y, R, C);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_transpose_bf16_batched(const uint16_t* x_base, uint16_t* y_base, const s2st_transpose_table& t, hipStream_t st) {
  if (t.n <= 0) return 0;
  if (t.n > S2ST_TRANSPOSE_MAX) return S2ST_ERR_ARG;
  S2ST_LAUNCH(transpose_bf16_batched_kernel, dim3(t.tile0[t.n]), dim3(256), 0, st, x_base, y_base, t);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_glu_fwd(const float* a, float* y, Split ysp, int rows, int C, hipStream_t st) {
  long n = (long)rows * C;
  if (n <= 0) return 0;
  S2ST_LAUNCH(glu_fwd_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, a, y, ysp, rows, C);
  return LAUNCH_OK();
}

int s2st_glu_bwd(const float* a, const float* dy, Split dysp, float* da, Split dasp, int rows,
                 int C, hipStream_t st, uint16_t* dah, long ldh) {
  long n = (long)rows * C;
  if (n <= 0) return 0;
  S2ST_LAUNCH(glu_bwd_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, a, dy, dysp, da, dasp, rows, C, dah, ldh);
  return LAUNCH_OK();
}

int s2st_add_pe(const float* x, float* y, const int* pos, const float* table, int rows, int C,
                float scale, const float* alpha_ptr, float drop_p, uint64_t seed, hipStream_t st,
                const float* spk_table, const long* spk_ids, int T) {
  long n = (long)rows * C;
  if (n <= 0) return 0;
  if (spk_table && (!spk_ids || T <= 0)) return S2ST_ERR_ARG;
  S2ST_LAUNCH(add_pe_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, y, pos, table, rows, C,
                     scale, alpha_ptr, drop_p, seed, spk_table, spk_ids, T > 0 ? T : 1);
  return LAUNCH_OK();
}

int s2st_speaker_bwd(const float* dy, const long* ids, int B, int T, int T_sum, int C, int n_spk, float drop_p,
                     uint64_t seed, float* dtable, hipStream_t st) {
  if (B <= 0 || C <= 0 || n_spk <= 0) return 0;
  S2ST_LAUNCH(speaker_bwd_kernel, dim3((C + 255) / 256, n_spk), dim3(256), 0, st, dy, ids, B, T, T_sum, C, n_spk, drop_p,
              seed, dtable);
  return LAUNCH_OK();
}

int s2st_speaker_fill_cols(const float* table, const long* ids, float* y, int B, int T, int ld, int coff, int Sd, hipStream_t st) {
  if (B <= 0 || T <= 0 || Sd <= 0) return 0;
  if (!table || !ids || !y || coff < 0 || coff + Sd > ld) return S2ST_ERR_ARG;
  S2ST_LAUNCH(speaker_fill_cols_kernel, dim3((unsigned)(((long)B * T * Sd + 255) / 256)), dim3(256), 0, st, table, ids, y, B, T,
              ld, coff, Sd);
  return LAUNCH_OK();
}

int s2st_speaker_cols_bwd(const float* dy, const long* ids, int B, int T, int ld, int coff, int Sd, int n_spk, float* dtable,
                          hipStream_t st) {
  if (B <= 0 || T <= 0 || Sd <= 0 || n_spk <= 0) return 0;
  if (!dy || !ids || !dtable || coff < 0 || coff + Sd > ld) return S2ST_ERR_ARG;
  S2ST_LAUNCH(speaker_cols_bwd_kernel, dim3((Sd + 255) / 256, n_spk), dim3(256), 0, st, dy, ids, B, T, ld, coff, Sd, n_spk, dtable);
  return LAUNCH_OK();
}

int s2st_split_cols(const float* g, int ldg, float* dx, int ldx, int rows, int C, int acc, hipStream_t st) {
  if (rows <= 0 || C <= 0) return 0;
  if (!g || !dx || C > ldg || C > ldx) return S2ST_ERR_ARG;
  S2ST_LAUNCH(split_cols_kernel, dim3((unsigned)(((long)rows * C + 255) / 256)), dim3(256), 0, st, g, ldg, dx, ldx, rows, C, acc);
  return LAUNCH_OK();
}

int s2st_speaker_set_rows(const float* table, const long* ids, float* y, int B, int T, int C, hipStream_t st) {
  if (B <= 0 || C <= 0) return 0;
  S2ST_LAUNCH(speaker_set_rows_kernel, dim3((unsigned)(((long)B * C + 255) / 256)), dim3(256), 0, st, table, ids, y, B, T, C);
  return LAUNCH_OK();
}

int s2st_pe_alpha_bwd(const float* dy, const int* pos, const float* table, int rows, int C,
                      float drop_p, uint64_t seed, float* dalpha, hipStream_t st, float* part, int* nparts_out) {
  long n = (long)rows * C;
  if (nparts_out) *nparts_out = 0;
  if (n <= 0) return 0;
  unsigned g = ew_grid(n, 8);
  if (g > 1024) g = 1024;
  if (!nparts_out) part = nullptr;
  S2ST_LAUNCH(pe_alpha_bwd_kernel, dim3(g), dim3(EW_BLOCK), 0, st, dy, pos, table, rows, C,
                     drop_p, seed, dalpha, part);
  if (part) *nparts_out = (int)g;
  return LAUNCH_OK();
}

int s2st_embed_fwd(const long* tokens, const float* table, float* y, int rows, int C, float scale,
                   hipStream_t st) {
  long n = (long)rows * C;
  if (n <= 0) return 0;
  S2ST_LAUNCH(embed_fwd_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, tokens, table, y, rows, C, scale);
  return LAUNCH_OK();
}

// dtable[v][c] += scale * sum over {r : tok[r] == v, v != pad} of dy[r][c], rows in index order (no atomics: the same
// bits every run).  One workgroup per token id: per window of EMB_WIN rows, every thread looks at EMB_WIN / 256
// consecutive rows, the matches are compacted IN ORDER into an LDS list (counts -> offsets -> rows), and the threads --
// one per column, EMB_CPT columns each -- add the listed rows.  Most ids have no row at all: those workgroups only scan.
// (First form: one thread per (id, column) walking all rows -- 0.17 ms per launch at 4 k rows x 1 k ids.)
constexpr int EMB_WIN = 4096, EMB_CPT = 4;
__global__ __launch_bounds__(256) void embed_bwd_ordered_kernel(const long* __restrict__ tok, const float* __restrict__ dy,
                                                                float* __restrict__ dtable, int rows, int C, int V,
                                                                float scale, long pad) {
  __shared__ int cnt[256];
  __shared__ int list[EMB_WIN];
  __shared__ int total;
  const int v = blockIdx.x, tid = threadIdx.x;
  if (v >= V || v == pad) return;
  float a[EMB_CPT] = {0.f, 0.f, 0.f, 0.f};
  bool any = false;
  constexpr int PT = EMB_WIN / 256;
  for (int w0 = 0; w0 < rows; w0 += EMB_WIN) {
    const int r0 = w0 + tid * PT;
    unsigned hit = 0;
#pragma unroll
    for (int q = 0; q < PT; ++q)
      if (r0 + q < rows && tok[r0 + q] == v) hit |= 1u << q;
    cnt[tid] = __builtin_popcount(hit);
    __syncthreads();
    int off = 0;
    for (int q = 0; q < tid; ++q) off += cnt[q];
    if (tid == 255) total = off + cnt[255];
    for (int q = 0; q < PT; ++q)
      if (hit >> q & 1) list[off++] = r0 + q;
    __syncthreads();
    const int n = total;
    if (n > 0) {
      any = true;
#pragma unroll
      for (int e = 0; e < EMB_CPT; ++e) {
        const int c = tid + 256 * e;
        if (c < C)
          for (int q = 0; q < n; ++q) a[e] += dy[(long)list[q] * C + c];
      }
    }
    __syncthreads();
  }
  if (any) {
#pragma unroll
    for (int e = 0; e < EMB_CPT; ++e) {
      const int c = tid + 256 * e;
      if (c < C) dtable[(long)v * C + c] += scale * a[e];
    }
  }
}

int s2st_embed_bwd(const long* tokens, const float* dy, float* dtable, int rows, int C, float scale,
                   long pad, hipStream_t st, int V) {
  long n = (long)rows * C;
  if (n <= 0) return 0;
  if (V > 0) {
    if (C > 256 * EMB_CPT) return S2ST_ERR_SHAPE;
    S2ST_LAUNCH(embed_bwd_ordered_kernel, dim3(V), dim3(256), 0, st, tokens, dy, dtable, rows, C, V, scale, pad);
    return LAUNCH_OK();
  }
  S2ST_LAUNCH(embed_bwd_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, tokens, dy, dtable, rows, C, scale, pad);
  return LAUNCH_OK();
}

int s2st_dropout(const float* x, float* y, long n, float a, float p, uint64_t seed, int accumulate,
                 hipStream_t st) {
  if (n <= 0) return 0;
  S2ST_LAUNCH(dropout_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, y, n, a, p, seed, accumulate);
  return LAUNCH_OK();
}

int s2st_relu_drop_bwd(const float* dy, const float* y, float* dz, long n, float p, hipStream_t st) {
  if (n <= 0) return 0;
  S2ST_LAUNCH(relu_drop_bwd_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, dy, y, dz, n,
                     p > 0.f ? 1.f / (1.f - p) : 1.f);
  return LAUNCH_OK();
}

static void dpre_geometry(int M, long ldp, int& cb, int& slabs, int& rpb) {
  cb = (int)((ldp + 255) / 256);
  slabs = (1024 + cb - 1) / cb;
  rpb = (M + slabs - 1) / slabs;
  if (rpb < 16) rpb = 16;
  slabs = (M + rpb - 1) / rpb;
}
long s2st_linear_bwd_prep_scratch_floats(int M, int N, long ldp) {
  if (M <= 0 || N <= 0) return 0;
  int cb, slabs, rpb;
  dpre_geometry(M, ldp, cb, slabs, rpb);
  return (long)slabs * N;
}

int s2st_linear_bwd_prep(const float* dy, const float* y, const uint16_t* yb, int mode, float p, uint64_t seed,
                         uint16_t* dph, long ldp, float* dpre, float* dbias, int M, int N, hipStream_t st, float* part,
                         int* slabs_out) {
  if (slabs_out) *slabs_out = 0;
  if (M <= 0 || N <= 0) return 0;
  if (N % 4 != 0 || ldp % 4 != 0 || ldp < N || ((uintptr_t)dy % 16) || (mode == 1 && !y && !yb) ||
      (mode == 1 && y && ((uintptr_t)y % 16)) || (mode == 1 && !y && ((uintptr_t)yb % 8)))
    return S2ST_ERR_SHAPE;
  int cb, slabs, rpb;
  dpre_geometry(M, ldp, cb, slabs, rpb);
  const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
  dim3 grid(cb, slabs);
  if (!dbias) part = nullptr;
  if (mode == 0) S2ST_LAUNCH(dpre_kernel<0>, grid, dim3(256), 0, st, dy, y, yb, dph, ldp, dpre, dbias, M, N, rpb, p, ik, seed, part);
  else if (mode == 1) S2ST_LAUNCH(dpre_kernel<1>, grid, dim3(256), 0, st, dy, y, yb, dph, ldp, dpre, dbias, M, N, rpb, p, ik, seed, part);
  else S2ST_LAUNCH(dpre_kernel<2>, grid, dim3(256), 0, st, dy, y, yb, dph, ldp, dpre, dbias, M, N, rpb, p, ik, seed, part);
  if (part && slabs_out) { *slabs_out = slabs; return LAUNCH_OK(); }
  if (part) return s2st_colsum_fold(part, slabs, N, dbias, st);
  return LAUNCH_OK();
}

int s2st_axpy(const float* x, float* y, long n, float a, hipStream_t st) {
  if (n <= 0) return 0;
  S2ST_LAUNCH(axpy_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, y, n, a);
  return LAUNCH_OK();
}

int s2st_scale(float* x, long n, float a, hipStream_t st) {
  if (n <= 0) return 0;
  S2ST_LAUNCH(scale_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, n, a);
  return LAUNCH_OK();
}

int s2st_conv_w_permute(const float* w, float* wf, float* wd, int O, int I, int Kw, hipStream_t st, uint16_t* wfh,
                        uint16_t* wdh) {
  long n = (long)O * I * Kw;
  if (n <= 0) return 0;
  if (O >= 32 && I >= 32 && Kw <= 64) {
    S2ST_LAUNCH(conv_w_permute_tiled_kernel, dim3((I + 31) / 32, (O + 31) / 32, Kw), dim3(256), 0, st, w, wf, wd, O, I, Kw,
                wfh, wdh);
    return LAUNCH_OK();
  }
  S2ST_LAUNCH(conv_w_permute_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, w, wf, wd, O, I, Kw, wfh, wdh);
  return LAUNCH_OK();
}

int s2st_conv_w_unpermute_acc(const float* dwf, float* dw, int O, int I, int Kw, hipStream_t st, int slabs) {
  long n = (long)O * I * Kw;
  if (n <= 0) return 0;
  S2ST_LAUNCH(conv_w_unpermute_acc_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, dwf, dw, O, I, Kw, slabs < 1 ? 1 : slabs);
  return LAUNCH_OK();
}

// tmp: S2ST_BN_TMP_FLOATS(C) floats of scratch
int s2st_bn_stats(const float* x, int rows, int C, float* mean, float* var, float* run_mean,
                  float* run_var, float momentum, float* tmp, hipStream_t st) {
  if (rows <= 0 || C <= 0) return 0;
  // one pass: shifted sums (d = x - x[row 0]; var = E[d^2] - E[d]^2), partials folded in slab order; 2 launches.  (Round 1 - 2's
  // mean pass + squared-deviation pass, 4 launches, was the A/B form of round 3: 7.82 vs 7.79 ms per step; gone.)
  int slabs = 0;
  int rc = colreduce2(ShiftSqF{x, C}, rows, C, nullptr, nullptr, tmp + 2 * (long)C, st, nullptr, nullptr, &slabs);
  if (rc) return rc;
  S2ST_LAUNCH(bn_finalize_shift_kernel, dim3((C + EW_BLOCK - 1) / EW_BLOCK), dim3(EW_BLOCK), 0, st, x,
              (const float*)(tmp + 2 * (long)C), slabs, mean, var, run_mean, run_var, C, rows, momentum);
  return LAUNCH_OK();
}

int s2st_bn_apply(const float* x, const float* mean, const float* var, const float* gamma,
                  const float* beta, float* y, Split ysp, const float* resid, int rows, int C,
                  float eps, int tanh_, float drop_p, uint64_t seed, hipStream_t st) {
  long n = (long)rows * C;
  if (n <= 0) return 0;
  S2ST_LAUNCH(bn_apply_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, mean, var, gamma,
                     beta, y, ysp, resid, rows, C, eps, tanh_, drop_p, seed);
  return LAUNCH_OK();
}

int s2st_bn_apply_img(const float* x, const float* mean, const float* var, const float* gamma, const float* beta, float* y,
                      uint16_t* img, int B, int T, int pad, int C, float eps, int tanh_, float drop_p, uint64_t seed,
                      hipStream_t st) {
  if (B <= 0 || T <= 0 || C <= 0) return 0;
  if (C % 4 || ((uintptr_t)x % 16) || ((uintptr_t)img % 8) || (y && (uintptr_t)y % 16)) return S2ST_ERR_SHAPE;
  const long n = (long)B * (T + 2 * pad) * (C / 4);
  S2ST_LAUNCH(bn_apply_img_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, mean, var, gamma, beta, y, img, B, T, pad, C / 4,
              eps, tanh_, drop_p, seed);
  return LAUNCH_OK();
}

int s2st_bn_bwd(const float* dy, Split dysp, const float* x, const float* mean, const float* var,
                const float* gamma, const float* beta, float* dx, Split dxsp, float* dgamma,
                float* dbeta, float* tmp, int rows, int C, float eps, int tanh_, float drop_p,
                uint64_t seed, hipStream_t st, uint16_t* dxh, long ldh) {
  long n = (long)rows * C;
  if (n <= 0) return 0;
  BnBwdF f{dy, dysp, x, mean, var, gamma, beta, C, eps, tanh_, drop_p, seed};
  // sums of dy' and dy' * xhat; the fold also adds them to the parameter gradients
  int rc = colreduce2(f, rows, C, tmp, tmp + C, tmp + 2 * (long)C, st, dbeta, dgamma);
  if (rc) return rc;
  S2ST_LAUNCH(bn_bwd_dx_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, f, tmp, dx, dxsp, rows, C, dxh, ldh);
  return LAUNCH_OK();
}
