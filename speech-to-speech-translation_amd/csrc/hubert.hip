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
__global__ __launch_bounds__(256) void hubert_conv0_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, int B, int N, int T, int C, int k,
                                                           int stride) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int cq = C >> 2;
  if (i >= (long)B * T * cq) return;
  const int c = (int)(i % cq) * 4;
  const long bt = i / cq;
  const int t = (int)(bt % T), b = (int)(bt / T);
  const float* xr = x + (long)b * N + (long)t * stride;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int j = 0; j < k; ++j) {
    const float xv = xr[j];
    a0 = fmaf(w[(c + 0) * k + j], xv, a0);
    a1 = fmaf(w[(c + 1) * k + j], xv, a1);
    a2 = fmaf(w[(c + 2) * k + j], xv, a2);
    a3 = fmaf(w[(c + 3) * k + j], xv, a3);
  }
  *reinterpret_cast<float4*>(y + bt * C + c) = make_float4(a0, a1, a2, a3);
}

// y = gelu(gamma[c] * (x - mean[b][c]) * rsqrt(var[b][c] + eps) + beta[c]) in place (+ bf16 copy)
__global__ __launch_bounds__(256) void gn_gelu_kernel(float* __restrict__ x, const float* __restrict__ mean,
                                                      const float* __restrict__ var, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, uint16_t* __restrict__ xh, int B,
                                                      int T, int C, float eps) {
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
    const float m = mean[(long)b * C + c + e], r = rsqrtf(var[(long)b * C + c + e] + eps);
    o[e] = gelu_erf((o[e] - m) * r * gamma[c + e] + beta[c + e]);
  }
  *reinterpret_cast<float4*>(x + bt * C + c) = make_float4(o[0], o[1], o[2], o[3]);
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

int s2st_hubert_conv0(const float* x, const float* w, float* y, int B, int N, int T, int C, int k, int stride,
                      hipStream_t st) {
  if (C % 4) return S2ST_ERR_SHAPE;
  const long n = (long)B * T * (C / 4);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(hubert_conv0_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, w, y, B, N, T, C, k,
                     stride);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_gn_gelu(float* x, const float* mean, const float* var, const float* gamma, const float* beta, uint16_t* xh,
                 int B, int T, int C, float eps, hipStream_t st) {
  if (C % 4) return S2ST_ERR_SHAPE;
  const long n = (long)B * T * (C / 4);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gn_gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, mean, var, gamma, beta, xh,
                     B, T, C, eps);
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
