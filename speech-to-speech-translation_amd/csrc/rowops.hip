// Row-wise kernels: LayerNorm fwd/bwd, attention softmax fwd/bwd (masks + dropout),
// column sums, head-mean of attention maps.  One wave64 per row, reductions by wave
// shuffles (no LDS on the row path), 16-byte loads when the row length allows.
//
// Reference call sites replaced: fairseq/modules/layer_norm.py:11-35 (F.layer_norm /
// apex FusedLayerNorm), fairseq/modules/multihead_attention.py:343-366 (mask fill,
// fp32 softmax, dropout) and the softmax inside F.multi_head_attention_forward (:170-192).
#include <cstdlib>

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

// LayerNorm backward in two independent kernels:
//   * layernorm_bwd_dx_kernel (data path): one wave per row, row statistics by shuffles, dx (= | +=) and, FUSE, the
//     bf16 GEMM operand dropout'(dx_total) of the linear layer that produced x.  No parameter-gradient work: the
//     earlier single kernel folded per-block column sums through LDS (a dozen barriers per launch) behind its row
//     pass and ran at 19 us for 4.6 k rows of 512 -- twice its memory time -- on the backward's critical path.
//   * layernorm_bwd_param_kernel (parameter gradients, off the critical path: the engine puts it on its second
//     stream): column sums over a slab of rows of dy * xhat (dgamma), dy (dbeta) and, FUSE, of the bf16 operand the
//     dx kernel wrote (bias gradient of the producing layer), one float4 column chunk per thread, no reductions
//     inside the slab; per-block partials are folded by layernorm_bwd_reduce_kernel in a fixed order.
template <bool FUSE>
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx, int dx_accumulate, int rows,
    int cols, uint16_t* __restrict__ dph, float drop_p, float inv_keep, uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;  // wave-uniform
  const int nv = cols >> 2;
  const float invc = 1.f / cols;
  const float* xr = x + (long)row * cols;
  const float* dyr = dy + (long)row * cols;
  float* dxr = dx + (long)row * cols;
  const float mu = mean[row], rs = rstd[row];
  float4 xh[LN_MAXV], g[LN_MAXV], old[LN_MAXV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      const float4 xv = reinterpret_cast<const float4*>(xr)[c4];
      const float4 dv = reinterpret_cast<const float4*>(dyr)[c4];
      const float4 gm = reinterpret_cast<const float4*>(gamma)[c4];
      if (dx_accumulate) old[i] = reinterpret_cast<const float4*>(dxr)[c4];
      xh[i].x = (xv.x - mu) * rs; xh[i].y = (xv.y - mu) * rs; xh[i].z = (xv.z - mu) * rs; xh[i].w = (xv.w - mu) * rs;
      g[i].x = dv.x * gm.x; g[i].y = dv.y * gm.y; g[i].z = dv.z * gm.z; g[i].w = dv.w * gm.w;
      s1 += g[i].x + g[i].y + g[i].z + g[i].w;
      s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
    }
  }
  s1 = wave_sum(s1) * invc;
  s2 = wave_sum(s2) * invc;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      float4 o;
      o.x = rs * (g[i].x - s1 - xh[i].x * s2);
      o.y = rs * (g[i].y - s1 - xh[i].y * s2);
      o.z = rs * (g[i].z - s1 - xh[i].z * s2);
      o.w = rs * (g[i].w - s1 - xh[i].w * s2);
      if (dx_accumulate) { o.x += old[i].x; o.y += old[i].y; o.z += old[i].z; o.w += old[i].w; }
      reinterpret_cast<float4*>(dxr)[c4] = o;
      if (FUSE) {
        const uint64_t e0 = (uint64_t)row * cols + 4 * c4;
        o.x *= drop_scale(seed, e0, drop_p, inv_keep); o.y *= drop_scale(seed, e0 + 1, drop_p, inv_keep);
        o.z *= drop_scale(seed, e0 + 2, drop_p, inv_keep); o.w *= drop_scale(seed, e0 + 3, drop_p, inv_keep);
        reinterpret_cast<uint2*>(dph + (long)row * cols)[c4] = pack_bf16x4(o.x, o.y, o.z, o.w);
      }
    }
  }
}

// grid (row slabs, column blocks of 256 float4 chunks); block 256 threads = lanes_c chunk lanes x rl row lanes.
// part: [gridDim.x][NOUT][cols]
template <bool FUSE>
__global__ __launch_bounds__(256) void layernorm_bwd_param_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const uint16_t* __restrict__ dph, float* __restrict__ part, int rows, int cols,
    int rows_per_block) {
  constexpr int NOUT = FUSE ? 3 : 2;
  __shared__ float4 red[NOUT][256];
  const int nv = cols >> 2;
  const int cb = blockIdx.y * 256;                       // first chunk of this column block
  const int lanes_c = min(nv - cb, 256);                 // chunk lanes in use
  const int rl = 256 / lanes_c;                          // row lanes
  const int cl = threadIdx.x % lanes_c, my_r = threadIdx.x / lanes_c;
  const int c4 = cb + cl;
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag, ad = ag;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  if (my_r < rl) {
    for (int r = r0 + my_r; r < r1; r += rl) {
      const float4 xv = reinterpret_cast<const float4*>(x + (long)r * cols)[c4];
      const float4 dv = reinterpret_cast<const float4*>(dy + (long)r * cols)[c4];
      const float mu = mean[r], rs = rstd[r];
      ag.x += dv.x * ((xv.x - mu) * rs); ag.y += dv.y * ((xv.y - mu) * rs);
      ag.z += dv.z * ((xv.z - mu) * rs); ag.w += dv.w * ((xv.w - mu) * rs);
      ab.x += dv.x; ab.y += dv.y; ab.z += dv.z; ab.w += dv.w;
      if (FUSE) {
        const uint2 h = reinterpret_cast<const uint2*>(dph + (long)r * cols)[c4];
        ad.x += __uint_as_float(h.x << 16); ad.y += __uint_as_float(h.x & 0xffff0000u);
        ad.z += __uint_as_float(h.y << 16); ad.w += __uint_as_float(h.y & 0xffff0000u);
      }
    }
  }
  // fold the row lanes (fixed order), one partial row per block
  red[0][threadIdx.x] = ag;
  red[1][threadIdx.x] = ab;
  if (FUSE) red[2][threadIdx.x] = ad;
  __syncthreads();
  if (my_r == 0) {
    for (int k = 1; k < rl; ++k) {
      const float4 a = red[0][k * lanes_c + cl], b = red[1][k * lanes_c + cl];
      ag.x += a.x; ag.y += a.y; ag.z += a.z; ag.w += a.w;
      ab.x += b.x; ab.y += b.y; ab.z += b.z; ab.w += b.w;
      if (FUSE) {
        const float4 d = red[2][k * lanes_c + cl];
        ad.x += d.x; ad.y += d.y; ad.z += d.z; ad.w += d.w;
      }
    }
    float* p = part + (long)blockIdx.x * NOUT * cols;
    reinterpret_cast<float4*>(p)[c4] = ag;
    reinterpret_cast<float4*>(p + cols)[c4] = ab;
    if (FUSE) reinterpret_cast<float4*>(p + 2 * cols)[c4] = ad;
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

// Round 3: ONE backward kernel on the data path.  The row pass above already holds dy, xhat and the rounded bf16 operand
// in registers; a wave that walks RPW rows keeps the column sums of its rows in registers too (a lane owns the same
// float4 column chunks in every row: no reduction inside the wave), the block's four waves meet once through 12 KB of
// LDS, and the block writes ONE partial row per output -- the slab partials the parameter kernel used to produce in a
// second pass over dy and x (18.8 MB per 4.6 k x 512 layer norm, 12 - 15 us on the second stream, 54 times a step).
// The partials of all layer norms of a backward segment are folded by one batched launch (fixed order: run-to-run
// identical sums).  All loads of a wave's RPW rows are issued before the first use.
template <bool FUSE, int NV, int RPW>
__global__ __launch_bounds__(256) void layernorm_bwd_fused_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx, int dx_accumulate, int rows,
    int cols, uint16_t* __restrict__ dph, float drop_p, float inv_keep, uint64_t seed, float* __restrict__ part) {
  constexpr int NOUT = FUSE ? 3 : 2;
  __shared__ float4 red[4][NOUT][64 * NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = cols >> 2;
  const float invc = 1.f / cols;
  const int row0 = (blockIdx.x * 4 + wave) * RPW;
  float4 gm[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    gm[i] = c4 < nv ? reinterpret_cast<const float4*>(gamma)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 xv[RPW][NV], dv[RPW][NV], old[RPW][NV];
  float mu[RPW], rs[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int row = min(row0 + r, rows - 1);  // clamped: loads stay unconditional; results of rows >= rows are dropped
    mu[r] = mean[row];
    rs[r] = rstd[row];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        xv[r][i] = reinterpret_cast<const float4*>(x + (long)row * cols)[c4];
        dv[r][i] = reinterpret_cast<const float4*>(dy + (long)row * cols)[c4];
        if (dx_accumulate) old[r][i] = reinterpret_cast<const float4*>(dx + (long)row * cols)[c4];
      }
    }
  }
  float4 ag[NV], ab[NV], ad[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) ag[i] = ab[i] = ad[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // Three straight-line phases over the wave's RPW rows (no control flow between rows, so the rows' shuffle reductions
  // interleave instead of forming one dependent chain per row): A) xhat, dy * gamma, the column sums and the two row
  // sums' lane partials; B) the 2 * RPW wave reductions; C) dx, the fused bf16 operand, stores.  Rows past the end
  // (clamped loads) are masked out of the column sums and not stored.
  float s1[RPW], s2[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const float live = row0 + r < rows ? 1.f : 0.f;
    s1[r] = s2[r] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        const float4 a = xv[r][i], d = dv[r][i];
        float4 xh, g;
        xh.x = (a.x - mu[r]) * rs[r]; xh.y = (a.y - mu[r]) * rs[r];
        xh.z = (a.z - mu[r]) * rs[r]; xh.w = (a.w - mu[r]) * rs[r];
        g.x = d.x * gm[i].x; g.y = d.y * gm[i].y; g.z = d.z * gm[i].z; g.w = d.w * gm[i].w;
        s1[r] += g.x + g.y + g.z + g.w;
        s2[r] += g.x * xh.x + g.y * xh.y + g.z * xh.z + g.w * xh.w;
        ag[i].x += live * d.x * xh.x; ag[i].y += live * d.y * xh.y; ag[i].z += live * d.z * xh.z; ag[i].w += live * d.w * xh.w;
        ab[i].x += live * d.x; ab[i].y += live * d.y; ab[i].z += live * d.z; ab[i].w += live * d.w;
        xv[r][i] = xh;  // (registers reused: xv now holds xhat, dv holds dy * gamma)
        dv[r][i] = g;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    s1[r] = wave_sum(s1[r]) * invc;
    s2[r] = wave_sum(s2[r]) * invc;
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int row = row0 + r;
    const bool live = row < rows;  // wave-uniform
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv && live) {
        const float4 xh = xv[r][i], g = dv[r][i];
        float4 o;
        o.x = rs[r] * (g.x - s1[r] - xh.x * s2[r]);
        o.y = rs[r] * (g.y - s1[r] - xh.y * s2[r]);
        o.z = rs[r] * (g.z - s1[r] - xh.z * s2[r]);
        o.w = rs[r] * (g.w - s1[r] - xh.w * s2[r]);
        if (dx_accumulate) { o.x += old[r][i].x; o.y += old[r][i].y; o.z += old[r][i].z; o.w += old[r][i].w; }
        reinterpret_cast<float4*>(dx + (long)row * cols)[c4] = o;
        if (FUSE) {
          const uint64_t e0 = (uint64_t)row * cols + 4 * c4;
          o.x *= drop_scale(seed, e0, drop_p, inv_keep); o.y *= drop_scale(seed, e0 + 1, drop_p, inv_keep);
          o.z *= drop_scale(seed, e0 + 2, drop_p, inv_keep); o.w *= drop_scale(seed, e0 + 3, drop_p, inv_keep);
          const uint2 h = pack_bf16x4(o.x, o.y, o.z, o.w);
          reinterpret_cast<uint2*>(dph + (long)row * cols)[c4] = h;
          // the bias gradient of the producing layer sums the ROUNDED operand (what its weight-gradient product sees)
          ad[i].x += __uint_as_float(h.x << 16); ad[i].y += __uint_as_float(h.x & 0xffff0000u);
          ad[i].z += __uint_as_float(h.y << 16); ad[i].w += __uint_as_float(h.y & 0xffff0000u);
        }
      }
    }
  }
  // the block's partial row of each output: every wave parks its column sums in LDS (one barrier), then the 256 threads
  // add the four waves' values in wave order, one float4 column chunk each (first form: three serial hand-offs to
  // wave 0, six barriers)
  float* p = part + (long)blockIdx.x * NOUT * cols;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    red[wave][0][lane + 64 * i] = ag[i];
    red[wave][1][lane + 64 * i] = ab[i];
    if (FUSE) red[wave][2][lane + 64 * i] = ad[i];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NOUT * 64 * NV; e += 256) {
    const int o = e / (64 * NV), c4 = e - o * (64 * NV);
    if (c4 >= nv) continue;
    float4 v = red[0][o][c4];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 t = red[w][o][c4];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    reinterpret_cast<float4*>(p + (long)o * cols)[c4] = v;
  }
}

// the fold of layernorm_bwd_reduce_kernel for a whole table of layer norms in one launch
__global__ __launch_bounds__(256) void layernorm_bwd_fold_batched_kernel(s2st_lnfold_table t) {
  __shared__ float red[8][32];
  int it = 0;
  while (it + 1 < t.n && (int)blockIdx.x >= t.blk0[it + 1]) ++it;
  const s2st_lnfold_item f = t.item[it];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = ((int)blockIdx.x - t.blk0[it]) * 32 + tx;  // over nout * cols
  const long ld = (long)f.nout * f.cols;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < f.nout * f.cols) {
    int b = ty;
    for (; b + 24 < f.nblocks; b += 32) {
      a0 += f.part[(long)b * ld + c];
      a1 += f.part[(long)(b + 8) * ld + c];
      a2 += f.part[(long)(b + 16) * ld + c];
      a3 += f.part[(long)(b + 24) * ld + c];
    }
    for (; b < f.nblocks; b += 8) a0 += f.part[(long)b * ld + c];
    if (f.part2) {  // the second chain's partial rows, behind the first chain's
      float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
      b = ty;
      for (; b + 24 < f.nblocks2; b += 32) {
        e0 += f.part2[(long)b * ld + c];
        e1 += f.part2[(long)(b + 8) * ld + c];
        e2 += f.part2[(long)(b + 16) * ld + c];
        e3 += f.part2[(long)(b + 24) * ld + c];
      }
      for (; b < f.nblocks2; b += 8) e0 += f.part2[(long)b * ld + c];
      a0 += e0; a1 += e1; a2 += e2; a3 += e3;
    }
  }
  red[ty][tx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ty == 0 && c < f.nout * f.cols) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) v += red[i][tx];
    if (c < f.cols) f.dgamma[c] += v;
    else if (c < 2 * f.cols) f.dbeta[c - f.cols] += v;
    else if (f.dbias) f.dbias[c - 2 * f.cols] += v;
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

// part != null: the slab's sums go to part[slab][cols] (folded in slab order by colsum_fold_kernel) instead of atomics
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, int rows,
                                                     int cols, float* __restrict__ out,
                                                     int rows_per_block, float* __restrict__ part) {
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
  if (ty == 0 && c < cols) {
    const float v = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    if (part) part[(long)blockIdx.y * cols + c] = v;
    else atomicAdd(&out[c], v);
  }
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

static int ln_param_rows_per_block(int rows) {
  // (split form) 32-row slabs (one partial row per block), at most 512 blocks
  int rpb = 32;
  while ((rows + rpb - 1) / rpb > 512) rpb *= 2;
  return rpb;
}
static int ln_split_blocks(int rows) {
  const int rpb = ln_param_rows_per_block(rows);
  const int b = (rows + rpb - 1) / rpb;
  return b < 1 ? 1 : b;
}
// fused form: 4 waves x RPW rows per block; RPW = 4 while a lane holds <= 2 float4 per row (cols <= 512), else 2
// (1 / 2 / 4 rows per wave measured within noise in round 3: the tuning switch is gone)
static int ln_rpw(int cols) { return cols <= 512 ? 4 : 2; }
static int ln_fused_rows_per_block(int cols) { return 4 * ln_rpw(cols); }
static int ln_fused_blocks(int rows, int cols) {
  const int rpb = ln_fused_rows_per_block(cols);
  const int b = (rows + rpb - 1) / rpb;
  return b < 1 ? 1 : b;
}

// partial rows either form writes (the scratch holds the larger of the two)
int s2st_layernorm_bwd_blocks(int rows, int cols) {
  const int a = ln_split_blocks(rows), b = ln_fused_blocks(rows, cols);
  return a > b ? a : b;
}

int s2st_layernorm_bwd_fold(const s2st_lnfold_table& t, hipStream_t st) {
  if (t.n <= 0) return 0;
  S2ST_LAUNCH(layernorm_bwd_fold_batched_kernel, dim3(t.blk0[t.n]), dim3(256), 0, st, t);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
int s2st_fold_add(s2st_lnfold_table& t, const float* part, int nblocks, int cols, int nout, float* out0, float* out1,
                  float* out2) {
  for (int j = 0; j < t.n; ++j) {
    s2st_lnfold_item& e = t.item[j];
    const bool shares = (out0 && (out0 == e.dgamma || out0 == e.dbeta || out0 == e.dbias)) ||
                        (out1 && (out1 == e.dgamma || out1 == e.dbeta || out1 == e.dbias)) ||
                        (out2 && (out2 == e.dgamma || out2 == e.dbeta || out2 == e.dbias));
    if (!shares) continue;
    // one output, two arrays of partial rows: the fold's "+=" is not atomic, so they become ONE entry
    if (e.dgamma != out0 || e.dbeta != out1 || e.dbias != out2 || e.cols != cols || e.nout != nout || e.part2) return S2ST_ERR_ARG;
    e.part2 = part; e.nblocks2 = nblocks;
    return 0;
  }
  if (t.n >= S2ST_LNFOLD_MAX) return S2ST_ERR_ARG;
  s2st_lnfold_item& f = t.item[t.n];
  f.part = part; f.dgamma = out0; f.dbeta = out1; f.dbias = out2;
  f.nblocks = nblocks; f.cols = cols; f.nout = nout; f.part2 = nullptr; f.nblocks2 = 0;
  t.blk0[t.n + 1] = t.blk0[t.n] + (nout * cols + 31) / 32;
  ++t.n;
  return 0;
}
int s2st_lnfold_add(s2st_lnfold_table& t, const float* part, int rows, int cols, int nout, float* dgamma, float* dbeta,
                    float* dbias) {
  return s2st_fold_add(t, part, ln_fused_blocks(rows, cols), cols, nout, dgamma, dbeta, dbias);
}

// scratch: s2st_layernorm_bwd_blocks(rows, cols) * (dph ? 3 : 2) * cols floats.
// dph != null: fused backward prologue of the linear layer that produced x (see the kernels): dph [rows][cols] bf16 =
// dropout'(dx_total) with the (seed, p) mask, dbias += its column sums (may be null).
// phase 0: everything on st (one row kernel that also leaves the column-sum partials + their fold);
//       3: that row kernel alone -- the caller folds the partials of several layer norms in one launch
//          (s2st_lnfold_add / s2st_layernorm_bwd_fold);
//    1, 2: the round-2 split form (1: the dx row kernel; 2: a second pass over dy and x for the parameter gradients +
//          its fold, which the engine used to put on its second stream) -- S2ST_LN_BWD_SPLIT=1, the A/B switch
int s2st_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                       const float* rstd, float* dx, int dx_accumulate, float* dgamma,
                       float* dbeta, float* scratch, int rows, int cols, hipStream_t st, int phase,
                       uint16_t* dph, float drop_p, uint64_t seed, float* dbias) {
  if (rows <= 0) return 0;
  if (cols % 4 != 0 || cols > LN_MAXV * 256) return S2ST_ERR_SHAPE;
  if (dph && ((uintptr_t)dph % 8 || drop_p < 0.f || drop_p >= 1.f)) return S2ST_ERR_ARG;
  if (phase == 0 || phase == 3) {
    const int blocks = ln_fused_blocks(rows, cols);
    // bytes: dy, x read; dx written (read too when it accumulates); the bf16 operand of the fused form
    const double by = (double)rows * cols * (12 + (dx_accumulate ? 4 : 0) + (dph ? 2 : 0));
    const float ik = dph ? 1.f / (1.f - drop_p) : 1.f;
#define LN_FUSED(FUSE, NV, RPW)                                                                                        \
    s2st_launch("layernorm_bwd_fused_kernel<" #FUSE ">", by, 0.0, layernorm_bwd_fused_kernel<FUSE, NV, RPW>, dim3(blocks), \
                dim3(256), 0, st, dy, x, gamma, mean, rstd, dx, dx_accumulate, rows, cols, dph, drop_p, ik, seed, scratch)
    const int rpw = ln_rpw(cols);
    if (rpw == 1) {
      if (cols <= 256) { if (dph) LN_FUSED(true, 1, 1); else LN_FUSED(false, 1, 1); }
      else if (cols <= 512) { if (dph) LN_FUSED(true, 2, 1); else LN_FUSED(false, 2, 1); }
      else { if (dph) LN_FUSED(true, 4, 1); else LN_FUSED(false, 4, 1); }
    } else
    if (cols <= 256) { if (rpw == 4) { if (dph) LN_FUSED(true, 1, 4); else LN_FUSED(false, 1, 4); } else { if (dph) LN_FUSED(true, 1, 2); else LN_FUSED(false, 1, 2); } }
    else if (cols <= 512) { if (rpw == 4) { if (dph) LN_FUSED(true, 2, 4); else LN_FUSED(false, 2, 4); } else { if (dph) LN_FUSED(true, 2, 2); else LN_FUSED(false, 2, 2); } }
    else { if (dph) LN_FUSED(true, 4, 2); else LN_FUSED(false, 4, 2); }
#undef LN_FUSED
    if (phase == 0) {
      s2st_lnfold_table t{};
      s2st_lnfold_add(t, scratch, rows, cols, dph ? 3 : 2, dgamma, dbeta, dbias);
      return s2st_layernorm_bwd_fold(t, st);
    }
    return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
  }
  if (phase != 2) {
    const double by = (double)rows * cols * (12 + (dx_accumulate ? 4 : 0) + (dph ? 2 : 0));
    if (dph)
      s2st_launch("layernorm_bwd_dx_kernel<true>", by, 0.0, layernorm_bwd_dx_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0,
                  st, dy, x, gamma, mean, rstd, dx, dx_accumulate, rows, cols, dph, drop_p, 1.f / (1.f - drop_p), seed);
    else
      s2st_launch("layernorm_bwd_dx_kernel<false>", by, 0.0, layernorm_bwd_dx_kernel<false>, dim3((rows + 3) / 4), dim3(256),
                  0, st, dy, x, gamma, mean, rstd, dx, dx_accumulate, rows, cols, (uint16_t*)nullptr, 0.f, 1.f, (uint64_t)0);
  }
  if (phase != 1) {
    const int rpb = ln_param_rows_per_block(rows), blocks = ln_split_blocks(rows);
    const int nout = dph ? 3 : 2;
    const dim3 grid(blocks, (cols / 4 + 255) / 256);
    const double by = (double)rows * cols * (8 + (dph ? 2 : 0));
    if (dph)
      s2st_launch("layernorm_bwd_param_kernel<true>", by, 0.0, layernorm_bwd_param_kernel<true>, grid, dim3(256), 0, st, dy, x,
                  mean, rstd, (const uint16_t*)dph, scratch, rows, cols, rpb);
    else
      s2st_launch("layernorm_bwd_param_kernel<false>", by, 0.0, layernorm_bwd_param_kernel<false>, grid, dim3(256), 0, st, dy,
                  x, mean, rstd, (const uint16_t*)nullptr, scratch, rows, cols, rpb);
    S2ST_LAUNCH(layernorm_bwd_reduce_kernel, dim3((nout * cols + 31) / 32), dim3(256), 0, st,
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
  S2ST_LAUNCH(softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, s, p,
                     pd, klen, B, H, T, S, ld, causal, drop_p, seed, pdh);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_softmax_bwd(const float* p, const float* dpd, float* ds, int B, int H, int T, int S,
                     int ld, float drop_p, uint64_t seed, hipStream_t st, uint16_t* dsh) {
  long rows = (long)B * H * T;
  if (rows <= 0) return 0;
  if (S > SM_MAXE * 64 || ld > SM_MAXE * 64) return S2ST_ERR_SHAPE;
  S2ST_LAUNCH(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p,
                     dpd, ds, rows, S, ld, drop_p, seed, dsh);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

__global__ void colsum_fold_kernel(const float* __restrict__ part, int slabs, int cols, float* __restrict__ out);
static void colsum_geometry(int rows, int cols, int& cb, int& slabs, int& rpb) {
  cb = (cols + 63) / 64;
  slabs = (512 + cb - 1) / cb;
  rpb = (rows + slabs - 1) / slabs;
  if (rpb < 16) rpb = 16;
  slabs = (rows + rpb - 1) / rpb;
}
long s2st_colsum_scratch_floats(int rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  int cb, slabs, rpb;
  colsum_geometry(rows, cols, cb, slabs, rpb);
  return (long)slabs * cols;
}
int s2st_colsum_fold(const float* part, int slabs, int cols, float* out, hipStream_t st) {
  if (slabs <= 0 || cols <= 0) return 0;
  S2ST_LAUNCH(colsum_fold_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, part, slabs, cols, out);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

// part (s2st_colsum_scratch_floats floats) != null: fixed-order sums (slab partials + fold) instead of fp32 atomics
int s2st_colsum(const float* x, long ld, int rows, int cols, float* out, int accumulate,
                hipStream_t st, float* part, int* slabs_out) {
  if (slabs_out) *slabs_out = 0;
  if (cols <= 0) return 0;
  if (!accumulate) hipMemsetAsync(out, 0, sizeof(float) * cols, st);
  if (rows <= 0) return 0;
  int cb, slabs, rpb;
  colsum_geometry(rows, cols, cb, slabs, rpb);
  S2ST_LAUNCH(colsum_kernel, dim3(cb, slabs), dim3(256), 0, st, x, ld, rows, cols, out, rpb, part);
  if (part && slabs_out) { *slabs_out = slabs; return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH; }
  if (part) return s2st_colsum_fold(part, slabs, cols, out, st);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

// The same sums WITHOUT atomics: slab partials [slabs][cols] into `part`, folded in slab order by a second small kernel.
// For column sums that are mathematically zero (the key projection's bias gradient: softmax is shift-invariant) the
// result is pure rounding noise, and Adam turns the noise's sign into a parameter step -- with atomics the noise depends
// on the order in which workgroups arrive, and a training run is not reproducible (Reproducibility).
__global__ __launch_bounds__(256) void colsum_bf16_part_kernel(const uint16_t* __restrict__ x, long ld, int rows, int cols,
                                                               float* __restrict__ part, int rows_per_block) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols)
    for (int rb = r0 + wave; rb < r1; rb += 16) {
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
    part[(long)blockIdx.y * cols + cc] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ part, int slabs, int cols,
                                                          float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int sb = 0; sb < slabs; sb += 8) {  // 8 loads in flight, adds in slab order
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = part[(long)min(sb + j, slabs - 1) * cols + c];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (sb + j < slabs) s += v[j];
  }
  out[c] += s;
}

static void colsum_bf16_geometry(int rows, int cols, int& cb, int& slabs, int& rpb) {
  cb = (cols + 255) / 256;
  slabs = (256 + cb - 1) / cb;  // ~256 workgroups: enough to stream, few enough partials to fold quickly
  rpb = (rows + slabs - 1) / slabs;
  if (rpb < 16) rpb = 16;
  slabs = (rows + rpb - 1) / rpb;
}

long s2st_colsum_bf16_scratch_floats(int rows, int cols) {
  int cb, slabs, rpb;
  colsum_bf16_geometry(rows, cols, cb, slabs, rpb);
  return (long)slabs * cols;
}

int s2st_colsum_bf16_ordered(const uint16_t* x, long ld, int rows, int cols, float* out, float* scratch, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  if (cols % 4 || ld % 4 || ((uintptr_t)x % 8) || !scratch) return S2ST_ERR_SHAPE;
  int cb, slabs, rpb;
  colsum_bf16_geometry(rows, cols, cb, slabs, rpb);
  S2ST_LAUNCH(colsum_bf16_part_kernel, dim3(cb, slabs), dim3(256), 0, st, x, ld, rows, cols, scratch, rpb);
  S2ST_LAUNCH(colsum_fold_kernel, dim3(cb), dim3(256), 0, st, (const float*)scratch, slabs, cols, out);
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
  S2ST_LAUNCH(colsum_bf16_kernel, dim3(cb, slabs), dim3(256), 0, st, x, ld, rows, cols, out, rpb);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_attn_headmean(const float* p, float* out, int B, int H, int T, int S, int ld,
                       hipStream_t st) {
  long n = (long)B * T * S;
  if (n <= 0) return 0;
  S2ST_LAUNCH(headmean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, out,
                     B, H, T, S, ld);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
