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
// the kernel runs over the waveform more than once and the conv output itself is never stored.  All passes evaluate the
// same FMA chain, so the statistics are those of exactly the values that get normalised.
// Round 4: TWO passes instead of three (rounds 1 - 3: sums, squared deviations, apply -- 207 + 184 + 383 us plus two 41 us
// folds for 24 x 8 s; now 124 + 13 + 302 us):
//   MODE 0: per (b, c) the sums of d and d^2 with d = conv - conv(frame 0) -- a data sample of the channel as the shift,
//           as the BatchNorm statistics do: var = E[d^2] - E[d]^2 loses a few ulp where the unshifted form can lose
//           everything --, one partial pair per time block, folded in block order by conv0_fold_kernel;
//   MODE 2: normalise + GELU + store (bf16 copy for the next conv's GEMM and / or fp32).
// and the waveform samples are read straight from global memory at wave-UNIFORM addresses (s_load_dwordx8 + x2 per frame
// into SGPRs that the FMAs take as operands) instead of 10 LDS broadcast reads per frame and thread: the statistics
// passes were bound by LDS-read issue, not by FMAs.  (Tried on the apply pass: a 12-instruction erf instead of libm's --
// no change, 302 us; round 5 again with the epilogues' branch-free erf AND eight channels per lane = one 1 KB store per wave
// and frame: 305 us -- neither the stores' width nor the erf; 0.63 GB of bf16 at 2.1 TB/s beside ~220 VALU instructions per
// frame and wave.)
// A workgroup owns C0_TT consecutive frames of one utterance and ALL channels; a thread keeps the taps of its two channels in
// registers and writes coalesced rows.
constexpr int C0_TT = 128, CONV0_MAXK = 16, CONV0_CPT = 2;  // frames per block, max taps, channels per thread
// KT: the number of taps the loops run over (k <= KT; with KT = CONV0_MAXK taps >= k carry zero weights and read clamped
// addresses): 10 for HuBERT / wav2vec 2.0's first block (k == 10 exactly), CONV0_MAXK otherwise -- compile-time so that a
// frame's sample loads are issued together, without a branch per tap
template <int MODE, int KT>
__global__ __launch_bounds__(256) void hubert_conv0_gn_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ mean_, const float* __restrict__ var_,
                                                              float* __restrict__ part, float* __restrict__ shift,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y,
                                                              uint16_t* __restrict__ yh, int B, int N, int T, int C,
                                                              int k, int stride, float eps) {
  constexpr bool GEN = KT == CONV0_MAXK;
  const int b = blockIdx.y, t0 = blockIdx.x * C0_TT, tid = threadIdx.x;
  const int nt = min(C0_TT, T - t0);
  const float* xu = x + (long)b * N;                 // the utterance (wave-uniform)
  const float* xb = xu + (long)t0 * stride;          // this block's first sample (wave-uniform)
  for (int c0 = tid * CONV0_CPT; c0 < C; c0 += 256 * CONV0_CPT) {
    float wr[CONV0_CPT][KT];
#pragma unroll
    for (int e = 0; e < CONV0_CPT; ++e)
#pragma unroll
      for (int j = 0; j < KT; ++j) wr[e][j] = (!GEN || j < k) ? w[(long)(c0 + e) * k + (GEN ? min(j, k - 1) : j)] : 0.f;
    float mu[CONV0_CPT] = {0.f, 0.f}, sc[CONV0_CPT] = {1.f, 1.f}, sh[CONV0_CPT] = {0.f, 0.f};
    if (MODE == 0) {  // the shift: the channel's value at the utterance's first frame
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        const float xv = xu[GEN ? min(j, k - 1) : j];
#pragma unroll
        for (int e = 0; e < CONV0_CPT; ++e) mu[e] = fmaf(wr[e][j], xv, mu[e]);
      }
      if (blockIdx.x == 0) {
#pragma unroll
        for (int e = 0; e < CONV0_CPT; ++e) shift[(long)b * C + c0 + e] = mu[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < CONV0_CPT; ++e) {
        mu[e] = mean_[(long)b * C + c0 + e];
        sc[e] = rsqrtf(var_[(long)b * C + c0 + e] + eps) * gamma[c0 + e];
        sh[e] = beta[c0 + e];
      }
    }
    float acc[CONV0_CPT] = {0.f, 0.f}, acq[CONV0_CPT] = {0.f, 0.f};
#pragma unroll 4
    for (int t = 0; t < nt; ++t) {
      float a[CONV0_CPT] = {0.f, 0.f};
      float xv[KT];
#pragma unroll
      for (int j = 0; j < KT; ++j) xv[j] = xb[t * stride + (GEN ? min(j, k - 1) : j)];  // (uniform addresses: scalar loads)
#pragma unroll
      for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int e = 0; e < CONV0_CPT; ++e) a[e] = fmaf(wr[e][j], xv[j], a[e]);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < CONV0_CPT; ++e) {
          const float d = a[e] - mu[e];
          acc[e] += d;
          acq[e] = fmaf(d, d, acq[e]);
        }
      } else {
        const float o0 = gelu_erf((a[0] - mu[0]) * sc[0] + sh[0]), o1 = gelu_erf((a[1] - mu[1]) * sc[1] + sh[1]);
        const long o = ((long)b * T + t0 + t) * C + c0;
        if (y) *reinterpret_cast<float2*>(y + o) = make_float2(o0, o1);
        // (nontemporal: 629 MB per 24 x 8 s of audio, read once by the next convolution -- more than the memory-side cache holds;
        //  300 -> 290 us and the next product 726 -> 714, config 3 -0.08 ms per step, profiles/r06_hubert_nontemporal_ab.txt)
        if (yh) __builtin_nontemporal_store(pack_bf16x4(o0, o1, 0.f, 0.f).x, reinterpret_cast<unsigned*>(yh + o));
      }
    }
    if (MODE == 0) {  // this block's share of the time sums: plain stores, folded in block order by conv0_fold_kernel
#pragma unroll
      for (int e = 0; e < CONV0_CPT; ++e) {
        part[(((long)b * gridDim.x + blockIdx.x) * 2 + 0) * C + c0 + e] = acc[e];
        part[(((long)b * gridDim.x + blockIdx.x) * 2 + 1) * C + c0 + e] = acq[e];
      }
    }
  }
}

// mean[b][c] = shift + (sum_blk d) / T, var[b][c] = (sum_blk d^2) / T - ((sum_blk d) / T)^2 (biased, as GroupNorm's), the
// partial sums added in block order (the same bits every run)
__global__ __launch_bounds__(256) void conv0_fold_kernel(const float* __restrict__ part, const float* __restrict__ shift,
                                                         float* __restrict__ mean, float* __restrict__ var, int B,
                                                         int nblk, int C, int T) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float s = 0.f, q = 0.f;
  for (int k0 = 0; k0 < nblk; k0 += 16) {  // (16 blocks' loads issued together; the adds stay in block order)
    float v0[16], v1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const long k = min(k0 + j, nblk - 1);
      v0[j] = part[((b * (long)nblk + k) * 2 + 0) * C + c];
      v1[j] = part[((b * (long)nblk + k) * 2 + 1) * C + c];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (k0 + j < nblk) {
        s += v0[j];
        q += v1[j];
      }
  }
  const float m = s / T;
  mean[i] = shift[i] + m;
  var[i] = fmaxf(q / T - m * m, 0.f);
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

long s2st_hubert_conv0_stats_floats(int B, int T, int C) { return (3 + 2 * (long)((T + C0_TT - 1) / C0_TT)) * B * C; }

// conv0 -> GroupNorm(C, C) over time -> GELU (see the kernel).  stats: scratch of s2st_hubert_conv0_stats_floats(B, T, C)
// floats (mean, variance, shift + two partial rows per time block: no atomics, nothing to zero); y (fp32) and yh
// (bf16), each optional, receive the [B][T][C] result.
int s2st_hubert_conv0_gn_gelu(const float* x, const float* w, const float* gamma, const float* beta, float* y,
                              uint16_t* yh, float* stats, int B, int N, int T, int C, int k, int stride, float eps,
                              hipStream_t st) {
  if (C % (2 * CONV0_CPT) || k > CONV0_MAXK || k < 1 || stride > 8 || stride < 1) return S2ST_ERR_SHAPE;
  if (B <= 0 || T <= 0) return 0;
  float *mean = stats, *var = stats + (long)B * C, *shift = stats + 2 * (long)B * C, *part = stats + 3 * (long)B * C;
  const int nblk = (T + C0_TT - 1) / C0_TT;
  const dim3 grid(nblk, B), fgrid((B * C + 255) / 256);
  const float *cm = mean, *cv = var;
  if (k == 10)
    S2ST_LAUNCH((hubert_conv0_gn_kernel<0, 10>), grid, dim3(256), 0, st, x, w, cm, cv, part, shift, gamma, beta, y, yh, B, N, T, C, k,
                stride, eps);
  else
    S2ST_LAUNCH((hubert_conv0_gn_kernel<0, CONV0_MAXK>), grid, dim3(256), 0, st, x, w, cm, cv, part, shift, gamma, beta, y, yh, B, N, T,
                C, k, stride, eps);
  S2ST_LAUNCH(conv0_fold_kernel, fgrid, dim3(256), 0, st, (const float*)part, (const float*)shift, mean, var, B, nblk, C, T);
  if (k == 10)
    S2ST_LAUNCH((hubert_conv0_gn_kernel<2, 10>), grid, dim3(256), 0, st, x, w, cm, cv, part, shift, gamma, beta, y, yh, B, N, T, C, k,
                stride, eps);
  else
    S2ST_LAUNCH((hubert_conv0_gn_kernel<2, CONV0_MAXK>), grid, dim3(256), 0, st, x, w, cm, cv, part, shift, gamma, beta, y, yh, B, N, T,
                C, k, stride, eps);
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
