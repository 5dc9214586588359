// Kernels of the frozen HuBERT front end (config 4) that are not GEMMs: the first waveform
// convolution (1 -> C channels), GroupNorm(C, C) + GELU over time, and the re-layout in front of the
// grouped positional convolution.  Activations are channel-last [B][T][C].
//
// Reference call sites replaced: fairseq/models/wav2vec/wav2vec2.py:777-783, 806-814
// (ConvFeatureExtractionModel, first block: Conv1d(1, C, k, stride) -> Fp32GroupNorm(C, C) -> GELU),
// :868-876 (index_put(x, padding_mask, 0) and the grouped pos_conv input).
#include "s2st_ops.h"

namespace {

// y[b][t][c] = sum_j w[c][j] * x[b][t * stride + j]     (no bias: conv_bias=False)
// A workgroup owns CONV0_TT consecutive frames of one utterance and ALL channels: the samples it needs
// sit in LDS (broadcast reads), a thread keeps the taps of its channels in registers and writes
// coalesced rows.  Fused: per-(b, c) sums of the outputs (first GroupNorm pass) via one atomic per
// channel per workgroup.
constexpr int CONV0_TT = 32, CONV0_MAXK = 16, CONV0_CPT = 2;  // frames per block, max taps, channels per thread
__global__ __launch_bounds__(256) void hubert_conv0_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, float* __restrict__ csum, int B,
                                                           int N, int T, int C, int k, int stride) {
  __shared__ float xs[CONV0_TT * 8 + CONV0_MAXK];
  const int b = blockIdx.y, t0 = blockIdx.x * CONV0_TT, tid = threadIdx.x;
  const int nt = min(CONV0_TT, T - t0);
  const int ns = (nt - 1) * stride + k;  // samples this block reads (stride <= 8 checked by the launcher)
  const float* xb = x + (long)b * N + (long)t0 * stride;
  for (int i = tid; i < ns; i += 256) xs[i] = xb[i];
  __syncthreads();
  for (int c0 = tid * CONV0_CPT; c0 < C; c0 += 256 * CONV0_CPT) {
    float wr[CONV0_CPT][CONV0_MAXK];
#pragma unroll
    for (int e = 0; e < CONV0_CPT; ++e)
#pragma unroll
      for (int j = 0; j < CONV0_MAXK; ++j) wr[e][j] = j < k ? w[(long)(c0 + e) * k + j] : 0.f;
    float sum[CONV0_CPT] = {0.f, 0.f};
    for (int t = 0; t < nt; ++t) {
      float a[CONV0_CPT] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < CONV0_MAXK; ++j) {
        if (j < k) {
          const float xv = xs[t * stride + j];
#pragma unroll
          for (int e = 0; e < CONV0_CPT; ++e) a[e] = fmaf(wr[e][j], xv, a[e]);
        }
      }
      *reinterpret_cast<float2*>(y + ((long)b * T + t0 + t) * C + c0) = make_float2(a[0], a[1]);
#pragma unroll
      for (int e = 0; e < CONV0_CPT; ++e) sum[e] += a[e];
    }
    if (csum) {
#pragma unroll
      for (int e = 0; e < CONV0_CPT; ++e) atomicAdd(csum + (long)b * C + c0 + e, sum[e]);
    }
  }
}

// second GroupNorm pass, all utterances in one launch: sq[b][c] += sum_t (x[b][t][c] - mean[b][c])^2
// (mean = csum / T).  grid (C / 256 * 4.., T slabs, B): a lane owns 4 channels, the 4 waves split the rows
__global__ __launch_bounds__(256) void gn_sqdev_kernel(const float* __restrict__ x, const float* __restrict__ csum,
                                                       float* __restrict__ sq, int T, int C, int rows_per_block) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.z;
  const int c = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(T, r0 + rows_per_block);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float4 mu4 = *reinterpret_cast<const float4*>(csum + (long)b * C + c);
    const float inv = 1.f / T;
    const float mu[4] = {mu4.x * inv, mu4.y * inv, mu4.z * inv, mu4.w * inv};
    for (int rb = r0 + wave; rb < r1; rb += 16) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = rb + 4 * u;
        v[u] = r < r1 ? *reinterpret_cast<const float4*>(x + ((long)b * T + r) * C + c) : make_float4(mu[0], mu[1], mu[2], mu[3]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d0 = v[u].x - mu[0], d1 = v[u].y - mu[1], d2 = v[u].z - mu[2], d3 = v[u].w - mu[3];
        a[0] += d0 * d0; a[1] += d1 * d1; a[2] += d2 * d2; a[3] += d3 * d3;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = a[e];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < C)
    atomicAdd(sq + (long)b * C + cc, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// y = gelu(gamma[c] * (x - mean[b][c]) * rsqrt(var[b][c] + eps) + beta[c]) in place (+ bf16 copy)
// mean / var are given as sums over the T frames (csum, sq); `write_f32` = 0 keeps only the bf16 copy
__global__ __launch_bounds__(256) void gn_gelu_kernel(float* __restrict__ x, const float* __restrict__ mean,
                                                      const float* __restrict__ var, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, uint16_t* __restrict__ xh, int B,
                                                      int T, int C, float eps, float inv_t, int write_f32) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int cq = C >> 2;
  if (i >= (long)B * T * cq) return;
  const int c = (int)(i % cq) * 4;
  const long bt = i / cq;
  const int b = (int)(bt / T);
  float4 v = *reinterpret_cast<float4*>(x + bt * C + c);
  float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float m = mean[(long)b * C + c + e] * inv_t, r = rsqrtf(var[(long)b * C + c + e] * inv_t + eps);
    o[e] = gelu_erf((o[e] - m) * r * gamma[c + e] + beta[c + e]);
  }
  if (write_f32) *reinterpret_cast<float4*>(x + bt * C + c) = make_float4(o[0], o[1], o[2], o[3]);
  if (xh) *reinterpret_cast<uint2*>(xh + bt * C + c) = pack_bf16x4(o[0], o[1], o[2], o[3]);
}

// x[b][t][:] = 0 for t >= lens[b] (in place), and the group-major, time-padded image
// img[g][b][pad + t][c] = x[b][t][g * Cg + c] (zeros in the pads), fp32 or bf16
__global__ __launch_bounds__(256) void posconv_prep_kernel(float* __restrict__ x, const int* __restrict__ lens,
                                                           float* __restrict__ img, uint16_t* __restrict__ imgh,
                                                           int B, int T, int E, int G, int pad, int Tp) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int eq = E >> 2;
  if (i >= (long)B * T * eq) return;
  const int e = (int)(i % eq) * 4;
  const long bt = i / eq;
  const int t = (int)(bt % T), b = (int)(bt / T);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < lens[b]) v = *reinterpret_cast<const float4*>(x + bt * E + e);
  else *reinterpret_cast<float4*>(x + bt * E + e) = v;
  const int Cg = E / G, g = e / Cg, c = e - g * Cg;  // Cg % 4 == 0
  const long o = (((long)g * B + b) * Tp + pad + t) * Cg + c;
  if (img) *reinterpret_cast<float4*>(img + o) = v;
  if (imgh) *reinterpret_cast<uint2*>(imgh + o) = pack_bf16x4(v.x, v.y, v.z, v.w);
}

}  // namespace

// conv0 + first GroupNorm pass: csum [B][C] (zeroed here) receives the per-utterance channel sums
int s2st_hubert_conv0(const float* x, const float* w, float* y, float* csum, int B, int N, int T, int C, int k,
                      int stride, hipStream_t st) {
  if (C % 4 || k > CONV0_MAXK || stride > 8 || stride < 1) return S2ST_ERR_SHAPE;  // C % 4: the GroupNorm kernels
  if (B <= 0 || T <= 0) return 0;
  if (csum) hipMemsetAsync(csum, 0, sizeof(float) * (size_t)B * C, st);
  hipLaunchKernelGGL(hubert_conv0_kernel, dim3((T + CONV0_TT - 1) / CONV0_TT, B), dim3(256), 0, st, x, w, y, csum, B, N,
                     T, C, k, stride);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

// GroupNorm(C, C) over the T frames of each utterance + GELU: csum from conv0, sq = scratch [B][C]
int s2st_gn_gelu(float* x, const float* csum, float* sq, const float* gamma, const float* beta, uint16_t* xh, int B,
                 int T, int C, float eps, int write_f32, hipStream_t st) {
  if (C % 4) return S2ST_ERR_SHAPE;
  const long n = (long)B * T * (C / 4);
  if (n <= 0) return 0;
  hipMemsetAsync(sq, 0, sizeof(float) * (size_t)B * C, st);
  const int cb = (C + 255) / 256;
  int slabs = (2048 / (cb * B)) > 0 ? 2048 / (cb * B) : 1;
  int rpb = (T + slabs - 1) / slabs;
  if (rpb < 16) rpb = 16;
  slabs = (T + rpb - 1) / rpb;
  hipLaunchKernelGGL(gn_sqdev_kernel, dim3(cb, slabs, B), dim3(256), 0, st, (const float*)x, csum, sq, T, C, rpb);
  hipLaunchKernelGGL(gn_gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, csum, (const float*)sq, gamma,
                     beta, xh, B, T, C, eps, 1.f / T, write_f32);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_posconv_prep(float* x, const int* lens, float* img, uint16_t* imgh, int B, int T, int E, int G, int pad,
                      int Tp, hipStream_t st) {
  if (E % G || (E / G) % 4) return S2ST_ERR_SHAPE;
  const long n = (long)B * T * (E / 4);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(posconv_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, lens, img, imgh, B, T,
                     E, G, pad, Tp);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
