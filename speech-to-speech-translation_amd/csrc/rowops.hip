// Row-wise kernels: LayerNorm fwd/bwd, attention softmax fwd/bwd (masks + dropout),
// column sums, head-mean of attention maps.  One wave64 per row, reductions by wave
// shuffles (no LDS on the row path), 16-byte loads when the row length allows.
//
// Reference call sites replaced: fairseq/modules/layer_norm.py:11-35 (F.layer_norm /
// apex FusedLayerNorm), fairseq/modules/multihead_attention.py:343-366 (mask fill,
// fp32 softmax, dropout) and the softmax inside F.multi_head_attention_forward (:170-192).
#include "s2st_ops.h"
#include "s2st_prof.h"

namespace {

constexpr int LN_MAXV = 4;  // float4 per lane cached in registers (cols <= 1024)

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ y,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out,
                                                            int rows, int cols, float eps,
                                                            uint16_t* __restrict__ yh) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;  // whole wave exits together (row is wave-uniform)
  const float* xr = x + (long)row * cols;
  float* yr = y + (long)row * cols;
  const int nv = cols >> 2;  // launcher guarantees cols % 4 == 0
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int c4 = lane + 64 * i;
    if (c4 < nv) {
      v[i] = reinterpret_cast<const float4*>(xr)[c4];
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
  }
  const float mean = wave_sum(s) / cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int c4 = lane + 64 * i;
    if (c4 < nv) {
      float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + c * c + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / cols + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int c4 = lane + 64 * i;
    if (c4 < nv) {
      float4 g = reinterpret_cast<const float4*>(gamma)[c4];
      float4 b = reinterpret_cast<const float4*>(beta)[c4];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (y) reinterpret_cast<float4*>(yr)[c4] = o;
      if (yh) reinterpret_cast<uint2*>(yh + (long)row * cols)[c4] = pack_bf16x4(o.x, o.y, o.z, o.w);
    }
  }
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

constexpr int LNB_WAVES = 8;  // 512-thread blocks: 2 waves per SIMD keep enough loads in flight (HBM-bound)

// FUSE: the input x of this layer norm is the output of a linear layer with dropout (x = resid + dropout(h W^T + b)),
// and this backward completes x's gradient: the kernel then also emits that layer's backward prologue -- the bf16
// GEMM operand dropout'(dx_total) (mask regenerated from (seed, element index)) and its column sums for the bias
// gradient (a third row of partials) -- instead of a separate pass over dx.
template <bool FUSE>
__global__ __launch_bounds__(64 * LNB_WAVES) void layernorm_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
    int dx_accumulate, float* __restrict__ part, int rows, int cols, uint16_t* __restrict__ dph, float drop_p,
    float inv_keep, uint64_t seed) {
  constexpr int NOUT = FUSE ? 3 : 2;
  __shared__ float red[2][LNB_WAVES][LN_MAXV * 64];  // [dgamma|dbeta][wave][float4 slot] (reused for the third row)
  float4 adb[FUSE ? LN_MAXV : 1];
#pragma unroll
  for (int i = 0; i < (FUSE ? LN_MAXV : 1); ++i) adb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = cols >> 2;
  const float invc = 1.f / cols;
  float4 ag[LN_MAXV], ab[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // gamma stays in registers; two rows per iteration (independent loads / shuffle chains overlap:
  // the loop is latency-bound at one row per wave)
  float4 gm[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int c4 = lane + 64 * i;
    gm[i] = c4 < nv ? reinterpret_cast<const float4*>(gamma)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int stride = gridDim.x * LNB_WAVES;
  for (int row0 = blockIdx.x * LNB_WAVES + wave; row0 < rows; row0 += 2 * stride) {
    float4 xh[2][LN_MAXV], g[2][LN_MAXV];
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, rs[2];
    bool on[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = row0 + u * stride;
      on[u] = row < rows;
      const int rr = on[u] ? row : row0;
      const float* xr = x + (long)rr * cols;
      const float* dyr = dy + (long)rr * cols;
      const float mu = mean[rr];
      rs[u] = rstd[rr];
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i) {
        int c4 = lane + 64 * i;
        if (c4 < nv) {
          float4 xv = reinterpret_cast<const float4*>(xr)[c4];
          float4 dv = reinterpret_cast<const float4*>(dyr)[c4];
          if (!on[u]) dv = make_float4(0.f, 0.f, 0.f, 0.f);
          xh[u][i].x = (xv.x - mu) * rs[u]; xh[u][i].y = (xv.y - mu) * rs[u];
          xh[u][i].z = (xv.z - mu) * rs[u]; xh[u][i].w = (xv.w - mu) * rs[u];
          g[u][i].x = dv.x * gm[i].x; g[u][i].y = dv.y * gm[i].y; g[u][i].z = dv.z * gm[i].z; g[u][i].w = dv.w * gm[i].w;
          s1[u] += g[u][i].x + g[u][i].y + g[u][i].z + g[u][i].w;
          s2[u] += g[u][i].x * xh[u][i].x + g[u][i].y * xh[u][i].y + g[u][i].z * xh[u][i].z + g[u][i].w * xh[u][i].w;
          ag[i].x += dv.x * xh[u][i].x; ag[i].y += dv.y * xh[u][i].y;
          ag[i].z += dv.z * xh[u][i].z; ag[i].w += dv.w * xh[u][i].w;
          ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
        }
      }
    }
    s1[0] = wave_sum(s1[0]) * invc; s1[1] = wave_sum(s1[1]) * invc;
    s2[0] = wave_sum(s2[0]) * invc; s2[1] = wave_sum(s2[1]) * invc;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!on[u]) continue;  // wave-uniform
      float* dxr = dx + (long)(row0 + u * stride) * cols;
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i) {
        int c4 = lane + 64 * i;
        if (c4 < nv) {
          float4 o;
          o.x = rs[u] * (g[u][i].x - s1[u] - xh[u][i].x * s2[u]);
          o.y = rs[u] * (g[u][i].y - s1[u] - xh[u][i].y * s2[u]);
          o.z = rs[u] * (g[u][i].z - s1[u] - xh[u][i].z * s2[u]);
          o.w = rs[u] * (g[u][i].w - s1[u] - xh[u][i].w * s2[u]);
          if (dx_accumulate) {
            float4 p = reinterpret_cast<float4*>(dxr)[c4];
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
          }
          reinterpret_cast<float4*>(dxr)[c4] = o;
          if (FUSE) {
            const uint64_t e0 = (uint64_t)(row0 + u * stride) * cols + 4 * c4;
            o.x *= drop_scale(seed, e0, drop_p, inv_keep); o.y *= drop_scale(seed, e0 + 1, drop_p, inv_keep);
            o.z *= drop_scale(seed, e0 + 2, drop_p, inv_keep); o.w *= drop_scale(seed, e0 + 3, drop_p, inv_keep);
            reinterpret_cast<uint2*>(dph + (long)(row0 + u * stride) * cols)[c4] = pack_bf16x4(o.x, o.y, o.z, o.w);
            adb[i].x += o.x; adb[i].y += o.y; adb[i].z += o.z; adb[i].w += o.w;
          }
        }
      }
    }
  }
  // cross-wave reduction of the parameter gradients, then one atomic per column per block
  float* rg = &red[0][0][0];
  float* rb = &red[1][0][0];
  constexpr int WSTRIDE = LN_MAXV * 64;  // one float4 component per pass
  for (int comp = 0; comp < 4; ++comp) {
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int c4 = lane + 64 * i;
      float gv = comp == 0 ? ag[i].x : comp == 1 ? ag[i].y : comp == 2 ? ag[i].z : ag[i].w;
      float bv = comp == 0 ? ab[i].x : comp == 1 ? ab[i].y : comp == 2 ? ab[i].z : ab[i].w;
      rg[wave * WSTRIDE + c4] = gv;
      rb[wave * WSTRIDE + c4] = bv;
    }
    __syncthreads();
    for (int c4 = threadIdx.x; c4 < nv; c4 += 64 * LNB_WAVES) {
      float gsum = 0.f, bsum = 0.f;
#pragma unroll
      for (int w = 0; w < LNB_WAVES; ++w) { gsum += rg[w * WSTRIDE + c4]; bsum += rb[w * WSTRIDE + c4]; }
      // per-block partials: [block][NOUT][cols]
      part[((long)blockIdx.x * NOUT + 0) * cols + c4 * 4 + comp] = gsum;
      part[((long)blockIdx.x * NOUT + 1) * cols + c4 * 4 + comp] = bsum;
    }
    __syncthreads();
    if (FUSE) {
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i) {
        int c4 = lane + 64 * i;
        rg[wave * WSTRIDE + c4] = comp == 0 ? adb[i].x : comp == 1 ? adb[i].y : comp == 2 ? adb[i].z : adb[i].w;
      }
      __syncthreads();
      for (int c4 = threadIdx.x; c4 < nv; c4 += 64 * LNB_WAVES) {
        float dsum = 0.f;
#pragma unroll
        for (int w = 0; w < LNB_WAVES; ++w) dsum += rg[w * WSTRIDE + c4];
        part[((long)blockIdx.x * NOUT + 2) * cols + c4 * 4 + comp] = dsum;
      }
      __syncthreads();
    }
  }
}

// out[c] += sum_b part[b][c] over the [nblocks][2*cols] partials (out = [dgamma | dbeta] halves).
// 256 threads = 32 columns x 8 row groups; a thread sums every 8th partial row with 4 loads in
// flight, the 8 groups are combined through LDS.
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* __restrict__ part,
                                                                   int nblocks, int cols,
                                                                   float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, int nout,
                                                                   float* __restrict__ dbias) {
  __shared__ float red[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;  // over nout * cols
  const long ld = (long)nout * cols;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < nout * cols) {
    int b = ty;
    for (; b + 24 < nblocks; b += 32) {
      a0 += part[(long)b * ld + c];
      a1 += part[(long)(b + 8) * ld + c];
      a2 += part[(long)(b + 16) * ld + c];
      a3 += part[(long)(b + 24) * ld + c];
    }
    for (; b < nblocks; b += 8) a0 += part[(long)b * ld + c];
  }
  red[ty][tx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ty == 0 && c < nout * cols) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) v += red[i][tx];
    if (c < cols) dgamma[c] += v;
    else if (c < 2 * cols) dbeta[c - cols] += v;
    else if (dbias) dbias[c - 2 * cols] += v;
  }
}

constexpr int SM_MAXE = 16;  // elements per lane cached (S <= 1024)

__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ s,
                                                          float* __restrict__ p,
                                                          float* __restrict__ pd,
                                                          const int* __restrict__ klen, int B,
                                                          int H, int T, int S, int ld, int causal,
                                                          float drop_p, uint64_t seed,
                                                          uint16_t* __restrict__ pdh) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long rows = (long)B * H * T;
  if (row >= rows) return;
  const int b = (int)(row / ((long)H * T));
  const int t = (int)(row % T);
  int lim = klen ? min(klen[b], S) : S;
  if (causal) lim = min(lim, t + 1);
  const float* sr = s + row * ld;
  float* pr = p + row * ld;
  float v[SM_MAXE];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    int c = lane + 64 * i;
    v[i] = (c < lim) ? sr[c] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    int c = lane + 64 * i;
    v[i] = (c < lim) ? expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  const float inv = 1.f / wave_sum(sum);
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    int c = lane + 64 * i;
    if (c < S) {
      float q = v[i] * inv;
      pr[c] = q;
      float qd = q;
      if (drop_p > 0.f) qd = q * drop_scale(seed, (uint64_t)row * ld + c, drop_p, inv_keep);
      if (pd) pd[row * ld + c] = qd;
      if (pdh) pdh[row * ld + c] = (uint16_t)(pack_bf16x4(qd, 0.f, 0.f, 0.f).x & 0xffffu);
    } else if (c < ld && pdh) {
      pdh[row * ld + c] = 0;
    }
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p,
                                                          const float* __restrict__ dpd,
                                                          float* __restrict__ ds, long rows, int S,
                                                          int ld, float drop_p, uint64_t seed,
                                                          uint16_t* __restrict__ dsh) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float pv[SM_MAXE], dv[SM_MAXE];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    int c = lane + 64 * i;
    pv[i] = dv[i] = 0.f;
    if (c < S) {
      pv[i] = p[row * ld + c];
      dv[i] = dpd[row * ld + c];
      if (drop_p > 0.f) dv[i] *= drop_scale(seed, (uint64_t)row * ld + c, drop_p, inv_keep);
      dot += pv[i] * dv[i];
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    int c = lane + 64 * i;
    if (c < S) {
      const float o = pv[i] * (dv[i] - dot);
      ds[row * ld + c] = o;
      if (dsh) dsh[row * ld + c] = (uint16_t)(pack_bf16x4(o, 0.f, 0.f, 0.f).x & 0xffffu);
    } else if (c < ld && dsh) {
      dsh[row * ld + c] = 0;
    }
  }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, int rows,
                                                     int cols, float* __restrict__ out,
                                                     int rows_per_block) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  float a = 0.f;
  if (c < cols)
    for (int r = r0 + ty; r < r1; r += 4) a += x[(long)r * ld + c];
  red[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && c < cols) atomicAdd(&out[c], red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}

// out[c] += sum_r bf16 x[r][c]: 4 columns per lane (8-byte loads), 4 waves over the rows of a slab
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const uint16_t* __restrict__ x, long ld, int rows,
                                                          int cols, float* __restrict__ out, int rows_per_block) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols)
    for (int rb = r0 + wave; rb < r1; rb += 16) {  // 4 rows per pass, loads first (latency-bound otherwise)
      uint2 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = rb + 4 * u;
        q[u] = r < r1 ? *reinterpret_cast<const uint2*>(x + (long)r * ld + c) : make_uint2(0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 += __uint_as_float(q[u].x << 16); a1 += __uint_as_float(q[u].x & 0xffff0000u);
        a2 += __uint_as_float(q[u].y << 16); a3 += __uint_as_float(q[u].y & 0xffff0000u);
      }
    }
  red[wave][lane * 4 + 0] = a0; red[wave][lane * 4 + 1] = a1;
  red[wave][lane * 4 + 2] = a2; red[wave][lane * 4 + 3] = a3;
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < cols)
    atomicAdd(out + cc, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void headmean_kernel(const float* __restrict__ p,
                                                       float* __restrict__ out, int B, int H, int T,
                                                       int S, int ld) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  long n = (long)B * T * S;
  if (i >= n) return;
  int s = (int)(i % S);
  int t = (int)((i / S) % T);
  int b = (int)(i / ((long)S * T));
  float a = 0.f;
  for (int h = 0; h < H; ++h) a += p[(((long)b * H + h) * T + t) * ld + s];
  out[((long)b * S + s) * T + t] = a / H;
}

}  // namespace

int s2st_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int rows, int cols, float eps, hipStream_t st,
                       uint16_t* yh) {
  if (rows <= 0) return 0;
  if (cols % 4 != 0 || cols > LN_MAXV * 256) return S2ST_ERR_SHAPE;
  // bytes: x read, y written in fp32 and / or bf16
  s2st_launch("layernorm_fwd_kernel", (double)rows * cols * (4 + (y ? 4 : 0) + (yh ? 2 : 0)), 0.0, layernorm_fwd_kernel,
              dim3((rows + 3) / 4), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, rows, cols, eps, yh);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_layernorm_bwd_blocks(int rows) {
  // two rows per wave in ONE pass when possible (the kernel is latency-bound: a second dependent
  // pass costs as much as the first), at most 512 blocks of partial sums
  int blocks = (rows + 2 * LNB_WAVES - 1) / (2 * LNB_WAVES);
  return blocks > 512 ? 512 : (blocks < 1 ? 1 : blocks);
}

// scratch: s2st_layernorm_bwd_blocks(rows) * (dph ? 3 : 2) * cols floats.
// dph != null: fused backward prologue of the linear layer that produced x (see the kernel): dph [rows][cols] bf16 =
// dropout'(dx_total) with the (seed, p) mask, dbias += its column sums (may be null).
int s2st_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                       const float* rstd, float* dx, int dx_accumulate, float* dgamma,
                       float* dbeta, float* scratch, int rows, int cols, hipStream_t st, int phase,
                       uint16_t* dph, float drop_p, uint64_t seed, float* dbias) {
  if (rows <= 0) return 0;
  if (cols % 4 != 0 || cols > LN_MAXV * 256) return S2ST_ERR_SHAPE;
  if (dph && ((uintptr_t)dph % 8 || drop_p < 0.f || drop_p >= 1.f)) return S2ST_ERR_ARG;
  int blocks = s2st_layernorm_bwd_blocks(rows);
  // phase 0: both kernels; 1: only dx + per-block partials; 2: only the parameter-gradient reduce (lets the
  // caller put that reduce on another stream: it is off the backward's critical path)
  if (phase != 2) {
    // bytes: dy, x read; dx written (read too when it accumulates); the bf16 operand of the fused form
    const double by = (double)rows * cols * (12 + (dx_accumulate ? 4 : 0) + (dph ? 2 : 0));
    if (dph)
      s2st_launch("layernorm_bwd_kernel<true>", by, 0.0, layernorm_bwd_kernel<true>, dim3(blocks), dim3(64 * LNB_WAVES), 0, st,
                  dy, x, gamma, mean, rstd, dx, dx_accumulate, scratch, rows, cols, dph, drop_p, 1.f / (1.f - drop_p), seed);
    else
      s2st_launch("layernorm_bwd_kernel<false>", by, 0.0, layernorm_bwd_kernel<false>, dim3(blocks), dim3(64 * LNB_WAVES), 0,
                  st, dy, x, gamma, mean, rstd, dx, dx_accumulate, scratch, rows, cols, (uint16_t*)nullptr, 0.f, 1.f,
                  (uint64_t)0);
  }
  if (phase != 1) {
    const int nout = dph ? 3 : 2;
    hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((nout * cols + 31) / 32), dim3(256), 0, st,
                       (const float*)scratch, blocks, cols, dgamma, dbeta, nout, dbias);
  }
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_softmax_fwd(const float* s, float* p, float* pd, const int* klen, int B, int H, int T,
                     int S, int ld, int causal, float drop_p, uint64_t seed, hipStream_t st,
                     uint16_t* pdh) {
  long rows = (long)B * H * T;
  if (rows <= 0) return 0;
  if (S > SM_MAXE * 64 || ld > SM_MAXE * 64) return S2ST_ERR_SHAPE;
  if (drop_p <= 0.f) pd = nullptr;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, s, p,
                     pd, klen, B, H, T, S, ld, causal, drop_p, seed, pdh);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_softmax_bwd(const float* p, const float* dpd, float* ds, int B, int H, int T, int S,
                     int ld, float drop_p, uint64_t seed, hipStream_t st, uint16_t* dsh) {
  long rows = (long)B * H * T;
  if (rows <= 0) return 0;
  if (S > SM_MAXE * 64 || ld > SM_MAXE * 64) return S2ST_ERR_SHAPE;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p,
                     dpd, ds, rows, S, ld, drop_p, seed, dsh);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_colsum(const float* x, long ld, int rows, int cols, float* out, int accumulate,
                hipStream_t st) {
  if (cols <= 0) return 0;
  if (!accumulate) hipMemsetAsync(out, 0, sizeof(float) * cols, st);
  if (rows <= 0) return 0;
  int cb = (cols + 63) / 64;
  int slabs = (512 + cb - 1) / cb;
  int rpb = (rows + slabs - 1) / slabs;
  if (rpb < 16) rpb = 16;
  slabs = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(colsum_kernel, dim3(cb, slabs), dim3(256), 0, st, x, ld, rows, cols, out, rpb);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_colsum_bf16(const uint16_t* x, long ld, int rows, int cols, float* out, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  if (cols % 4 || ld % 4 || ((uintptr_t)x % 8)) return S2ST_ERR_SHAPE;
  const int cb = (cols + 255) / 256;
  int slabs = (1024 + cb - 1) / cb;
  int rpb = (rows + slabs - 1) / slabs;
  if (rpb < 16) rpb = 16;
  slabs = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cb, slabs), dim3(256), 0, st, x, ld, rows, cols, out, rpb);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_attn_headmean(const float* p, float* out, int B, int H, int T, int S, int ld,
                       hipStream_t st) {
  long n = (long)B * T * S;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(headmean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, out,
                     B, H, T, S, ld);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
