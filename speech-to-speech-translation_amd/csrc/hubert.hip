// Kernels of the frozen HuBERT front end (config 4) that are not GEMMs: the first waveform
// convolution (1 -> C channels), GroupNorm(C, C) + GELU over time, and the re-layout in front of the
// grouped positional convolution.  Activations are channel-last [B][T][C].
//
// Reference call sites replaced: fairseq/models/wav2vec/wav2vec2.py:777-783, 806-814
// (ConvFeatureExtractionModel, first block: Conv1d(1, C, k, stride) -> Fp32GroupNorm(C, C) -> GELU),
// :868-876 (index_put(x, padding_mask, 0) and the grouped pos_conv input).
#include "s2st_ops.h"
#include "s2st_prof.h"

namespace {

// First block of the feature extractor, fused: y[b][t][c] = gelu(GroupNorm_c(conv0)[b][t][c]) with
// conv0[b][t][c] = sum_j w[c][j] * x[b][t * stride + j] (no bias: conv_bias=False) and GroupNorm(C, C) statistics
// over the T frames of each (utterance, channel).
// The convolution has ONE input channel and k = 10 taps: recomputing it is ~30x cheaper than a round trip of its
// [B][T][C] fp32 output through HBM (1.26 GB for 24 x 8 s of audio), and the waveform (12 MB) stays in L2.  So
// the kernel runs three times over the waveform -- MODE 0: per-(b, c) sums; MODE 1: squared deviations from the
// mean; MODE 2: normalise + GELU + store (bf16 copy for the next conv's GEMM and / or fp32) -- and the conv
// output itself is never stored.  The three passes evaluate the same FMA chain, so the statistics are those of
// exactly the values that get normalised.
// A workgroup owns C0_TT consecutive frames of one utterance and ALL channels: the samples it needs sit in LDS
// (broadcast reads), a thread keeps the taps of its two channels in registers and writes coalesced rows.
constexpr int C0_TT = 128, CONV0_MAXK = 16, CONV0_CPT = 2;  // frames per block, max taps, channels per thread
template <int MODE>
__global__ __launch_bounds__(256) void hubert_conv0_gn_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ csum, float* __restrict__ sq,
                                                              float* __restrict__ part,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y,
                                                              uint16_t* __restrict__ yh, int B, int N, int T, int C,
                                                              int k, int stride, float eps) {
  __shared__ float xs[C0_TT * 8 + CONV0_MAXK];
  const int b = blockIdx.y, t0 = blockIdx.x * C0_TT, tid = threadIdx.x;
  const int nt = min(C0_TT, T - t0);
  const int ns = (nt - 1) * stride + k;  // samples this block reads (stride <= 8 checked by the launcher)
  const float* xb = x + (long)b * N + (long)t0 * stride;
  for (int i = tid; i < ns; i += 256) xs[i] = xb[i];
  __syncthreads();
  const float inv_t = 1.f / T;
  for (int c0 = tid * CONV0_CPT; c0 < C; c0 += 256 * CONV0_CPT) {
    float wr[CONV0_CPT][CONV0_MAXK];
#pragma unroll
    for (int e = 0; e < CONV0_CPT; ++e)
#pragma unroll
      for (int j = 0; j < CONV0_MAXK; ++j) wr[e][j] = j < k ? w[(long)(c0 + e) * k + j] : 0.f;
    float mu[CONV0_CPT] = {0.f, 0.f}, sc[CONV0_CPT] = {1.f, 1.f}, sh[CONV0_CPT] = {0.f, 0.f};
    if (MODE >= 1) {
#pragma unroll
      for (int e = 0; e < CONV0_CPT; ++e) mu[e] = csum[(long)b * C + c0 + e] * inv_t;
    }
    if (MODE == 2) {
#pragma unroll
      for (int e = 0; e < CONV0_CPT; ++e) {
        sc[e] = rsqrtf(sq[(long)b * C + c0 + e] * inv_t + eps) * gamma[c0 + e];
        sh[e] = beta[c0 + e];
      }
    }
    float acc[CONV0_CPT] = {0.f, 0.f};
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
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < CONV0_CPT; ++e) acc[e] += a[e];
      } else if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < CONV0_CPT; ++e) { const float d = a[e] - mu[e]; acc[e] += d * d; }
      } else {
        const float o0 = gelu_erf((a[0] - mu[0]) * sc[0] + sh[0]), o1 = gelu_erf((a[1] - mu[1]) * sc[1] + sh[1]);
        const long o = ((long)b * T + t0 + t) * C + c0;
        if (y) *reinterpret_cast<float2*>(y + o) = make_float2(o0, o1);
        if (yh) *reinterpret_cast<unsigned*>(yh + o) = pack_bf16x4(o0, o1, 0.f, 0.f).x;
      }
    }
    if (MODE < 2) {  // this block's share of the time sums: a plain store, folded in block order by conv0_fold_kernel
#pragma unroll
      for (int e = 0; e < CONV0_CPT; ++e) part[((long)b * gridDim.x + blockIdx.x) * C + c0 + e] = acc[e];
    }
  }
}

// out[b][c] = sum over the nblk time blocks of part[b][blk][c], in block order (the same bits every run)
__global__ __launch_bounds__(256) void conv0_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int B,
                                                         int nblk, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float s = 0.f;
  for (int k = 0; k < nblk; ++k) s += part[((long)b * nblk + k) * C + c];
  out[i] = s;
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

long s2st_hubert_conv0_stats_floats(int B, int T, int C) { return (2 + (long)((T + C0_TT - 1) / C0_TT)) * B * C; }

// conv0 -> GroupNorm(C, C) over time -> GELU (see the kernel).  stats: scratch of s2st_hubert_conv0_stats_floats(B, T, C)
// floats (the two statistics + one partial row per time block: no atomics, nothing to zero); y (fp32) and yh
// (bf16), each optional, receive the [B][T][C] result.
int s2st_hubert_conv0_gn_gelu(const float* x, const float* w, const float* gamma, const float* beta, float* y,
                              uint16_t* yh, float* stats, int B, int N, int T, int C, int k, int stride, float eps,
                              hipStream_t st) {
  if (C % (2 * CONV0_CPT) || k > CONV0_MAXK || stride > 8 || stride < 1) return S2ST_ERR_SHAPE;
  if (B <= 0 || T <= 0) return 0;
  float *csum = stats, *sq = stats + (long)B * C, *part = stats + 2 * (long)B * C;
  const int nblk = (T + C0_TT - 1) / C0_TT;
  const dim3 grid(nblk, B), fgrid((B * C + 255) / 256);
  S2ST_LAUNCH(hubert_conv0_gn_kernel<0>, grid, dim3(256), 0, st, x, w, csum, sq, part, gamma, beta, y, yh, B, N, T, C,
                     k, stride, eps);
  S2ST_LAUNCH(conv0_fold_kernel, fgrid, dim3(256), 0, st, part, csum, B, nblk, C);
  S2ST_LAUNCH(hubert_conv0_gn_kernel<1>, grid, dim3(256), 0, st, x, w, csum, sq, part, gamma, beta, y, yh, B, N, T, C,
                     k, stride, eps);
  S2ST_LAUNCH(conv0_fold_kernel, fgrid, dim3(256), 0, st, part, sq, B, nblk, C);
  S2ST_LAUNCH(hubert_conv0_gn_kernel<2>, grid, dim3(256), 0, st, x, w, csum, sq, part, gamma, beta, y, yh, B, N, T, C,
                     k, stride, eps);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_posconv_prep(float* x, const int* lens, float* img, uint16_t* imgh, int B, int T, int E, int G, int pad,
                      int Tp, hipStream_t st) {
  if (E % G || (E / G) % 4) return S2ST_ERR_SHAPE;
  const long n = (long)B * T * (E / 4);
  if (n <= 0) return 0;
  S2ST_LAUNCH(posconv_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, lens, img, imgh, B, T,
                     E, G, pad, Tp);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
