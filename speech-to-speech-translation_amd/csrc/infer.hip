// Inference-only kernels of the AR mel decoder and the Griffin-Lim vocoder (config 5).
//
// Reference call sites replaced: fairseq/modules/multihead_attention.py:194-385 with
// incremental_state (one query per utterance against the cached keys / values, key padding mask,
// head-averaged weights of the alignment layer), fairseq/speech_generator_for_s2st.py:88-110
// (sigmoid of the stop logits, argmax alignment, global-CMVN de-normalisation) and
// fairseq/models/text_to_speech/vocoder.py:84-110 + fairseq/data/audio/audio_utils.py:259-271
// (Griffin-Lim: polar <-> rectangular spectra around the dense-DFT GEMMs, overlap-add).
#include "s2st_ops.h"

namespace {

constexpr int DA_MAXS = 4096;

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// one query per (b, h): o[b][h*dh + d] = sum_s softmax_s(scale q.k_s) v_s[d]; optional
// attn_mean[b][s] += p_s / H (zero-initialised by the launcher)
__global__ __launch_bounds__(256) void decode_attn_kernel(const float* __restrict__ q, long ldq,
                                                          const float* __restrict__ kc, const float* __restrict__ vc,
                                                          long ldk, long kbs, const int* __restrict__ klen, int nkeys,
                                                          int H, int dh, float scale, float* __restrict__ o, long ldo,
                                                          float* __restrict__ attn_mean, int S) {
  __shared__ float p[DA_MAXS];
  __shared__ float qs[256];
  __shared__ float red[4];
  __shared__ float part[256];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, tid = threadIdx.x;
  const int n = klen ? min((int)klen[b], nkeys) : nkeys;
  if (tid < dh) qs[tid] = q[(long)b * ldq + h * dh + tid] * scale;
  __syncthreads();
  const float* kb = kc + (long)b * kbs + h * dh;
  const float* vb = vc + (long)b * kbs + h * dh;
  float mx = -INFINITY;
  for (int s = tid; s < n; s += 256) {
    const float* kr = kb + (long)s * ldk;
    float a = 0.f;
    for (int d = 0; d < dh; d += 4) {
      const float4 kv = *reinterpret_cast<const float4*>(kr + d);
      a += qs[d] * kv.x + qs[d + 1] * kv.y + qs[d + 2] * kv.z + qs[d + 3] * kv.w;
    }
    p[s] = a;
    mx = fmaxf(mx, a);
  }
  mx = block_max(mx, red);
  float sum = 0.f;
  for (int s = tid; s < n; s += 256) {
    const float e = __expf(p[s] - mx);
    p[s] = e;
    sum += e;
  }
  sum = block_sum(sum, red);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  __syncthreads();
  // o: thread -> (d = tid % dh, key slice tid / dh)
  const int slices = 256 / dh, d = tid % dh, sl = tid / dh;
  float acc = 0.f;
  if (sl < slices)
    for (int s = sl; s < n; s += slices) acc += p[s] * vb[(long)s * ldk + d];
  part[tid] = acc;
  __syncthreads();
  if (tid < dh) {
    float a = 0.f;
    for (int i = 0; i < slices; ++i) a += part[i * dh + tid];
    o[(long)b * ldo + h * dh + tid] = a * inv;
  }
  if (attn_mean)
    for (int s = tid; s < n; s += 256) atomicAdd(attn_mean + (long)b * S + s, p[s] * inv / H);
}

__global__ __launch_bounds__(256) void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = 1.f / (1.f + __expf(-x[i]));
}

// x [B][E][D] -> idx[b][d] = first argmax_e x[b][e][d]
__global__ __launch_bounds__(256) void argmax_dim1_kernel(const float* __restrict__ x, long* __restrict__ idx, int B,
                                                          int E, int D) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * D) return;
  const int b = (int)(i / D), d = (int)(i - (long)b * D);
  const float* xb = x + (long)b * E * D + d;
  float best = xb[0];
  int bi = 0;
  for (int e = 1; e < E; ++e) {
    const float v = xb[(long)e * D];
    if (v > best) { best = v; bi = e; }
  }
  idx[i] = bi;
}

// y[r][c] = x[r][c] * scale[c] + shift[c]
__global__ __launch_bounds__(256) void affine_cols_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                          const float* __restrict__ sh, float* __restrict__ y,
                                                          long rows, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C) return;
  const int c = (int)(i % C);
  y[i] = x[i] * sc[c] + sh[c];
}

// y[c][t] = exp(x[t][c])   (vocoder.py:139: x.exp().transpose(-1, -2))
__global__ __launch_bounds__(256) void exp_transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int T,
                                                            int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)T * C) return;
  const int c = (int)(i / T), t = (int)(i - (long)c * T);
  y[i] = expf(x[(long)t * C + c]);
}
__global__ __launch_bounds__(256) void clamp_min_kernel(float* __restrict__ x, long n, float lo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = fmaxf(x[i], lo);
}

// ---- Griffin-Lim pieces.  Spectra are kept frame-major: X[t][0..F-1] = real, X[t][F..2F-1] = imag ----
// X = mag * (cos, sin)(angle);  mag [F][T] (reference layout), angle [F][T]
__global__ __launch_bounds__(256) void gl_polar_kernel(const float* __restrict__ mag, const float* __restrict__ ang,
                                                       float* __restrict__ X, int F, int T) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)F * T) return;
  const int f = (int)(i / T), t = (int)(i - (long)f * T);
  const float m = mag[i], a = ang[i];
  X[(long)t * 2 * F + f] = m * cosf(a);
  X[(long)t * 2 * F + F + f] = m * sinf(a);
}
// X = mag * (cos, sin)(atan2(Yi, Yr));  Y [T][2F] is the STFT of the current waveform
__global__ __launch_bounds__(256) void gl_project_kernel(const float* __restrict__ mag, const float* __restrict__ Y,
                                                         float* __restrict__ X, int F, int T) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)F * T) return;
  const int t = (int)(i / F), f = (int)(i - (long)t * F);
  const float re = Y[(long)t * 2 * F + f], im = Y[(long)t * 2 * F + F + f];
  const float a = atan2f(im, re), m = mag[(long)f * T + t];
  X[(long)t * 2 * F + f] = m * cosf(a);
  X[(long)t * 2 * F + F + f] = m * sinf(a);
}
// reflect-pad a waveform by `pad` on both sides: y[i] = x[reflect(i - pad)]
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                          int pad) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)n + 2 * pad) return;
  int j = (int)i - pad;
  if (j < 0) j = -j;
  if (j >= n) j = 2 * (n - 1) - j;
  y[i] = x[j];
}
// overlap-add of frames [T][n_fft] at hop, / window-sum-square (where > tiny), * n_fft/hop, trimmed by
// n_fft/2 at both ends: wave[i], i in [0, hop*(T-1))
__global__ __launch_bounds__(256) void gl_overlap_add_kernel(const float* __restrict__ frames,
                                                             const float* __restrict__ wsq, float* __restrict__ wave,
                                                             int T, int n_fft, int hop, int n_out, float tiny) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const int pos = (int)i + n_fft / 2;  // position in the untrimmed signal
  int t1 = pos / hop;
  if (t1 > T - 1) t1 = T - 1;
  int t0;  // first frame covering pos: t*hop + n_fft > pos
  if (pos - n_fft + 1 <= 0) t0 = 0;
  else t0 = (pos - n_fft + 1 + hop - 1) / hop;
  float a = 0.f;
  for (int t = t0; t <= t1; ++t) a += frames[(long)t * n_fft + (pos - t * hop)];
  const float w = wsq[pos];
  if (w > tiny) a /= w;
  wave[i] = a * ((float)n_fft / (float)hop);
}

}  // namespace

int s2st_decode_attn(const float* q, long ldq, const float* kc, const float* vc, long ldk, long kbs, const int* klen,
                     int nkeys, int B, int H, int dh, float scale, float* o, long ldo, float* attn_mean, int S,
                     hipStream_t st) {
  if (B <= 0) return 0;
  if (nkeys > DA_MAXS || dh > 256 || dh % 4 || 256 % dh) return S2ST_ERR_SHAPE;
  if (attn_mean) hipMemsetAsync(attn_mean, 0, sizeof(float) * (size_t)B * S, st);
  hipLaunchKernelGGL(decode_attn_kernel, dim3(B * H), dim3(256), 0, st, q, ldq, kc, vc, ldk, kbs, klen, nkeys, H, dh,
                     scale, o, ldo, attn_mean, S);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_sigmoid(const float* x, float* y, long n, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_argmax_dim1(const float* x, long* idx, int B, int E, int D, hipStream_t st) {
  const long n = (long)B * D;
  if (n <= 0 || E <= 0) return 0;
  hipLaunchKernelGGL(argmax_dim1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, idx, B, E, D);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_affine_cols(const float* x, const float* scale, const float* shift, float* y, long rows, int C,
                     hipStream_t st) {
  const long n = rows * C;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(affine_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, scale, shift, y, rows,
                     C);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_exp_transpose(const float* x, float* y, int T, int C, hipStream_t st) {
  const long n = (long)T * C;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(exp_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, T, C);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
int s2st_clamp_min(float* x, long n, float lo, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(clamp_min_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, lo);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_gl_polar(const float* mag, const float* ang, float* X, int F, int T, hipStream_t st) {
  const long n = (long)F * T;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gl_polar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mag, ang, X, F, T);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_gl_project(const float* mag, const float* Y, float* X, int F, int T, hipStream_t st) {
  const long n = (long)F * T;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gl_project_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mag, Y, X, F, T);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_reflect_pad(const float* x, float* y, int n, int pad, hipStream_t st) {
  if (n <= pad) return S2ST_ERR_SHAPE;
  const long m = (long)n + 2 * pad;
  hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, x, y, n, pad);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_gl_overlap_add(const float* frames, const float* wsq, float* wave, int T, int n_fft, int hop, int n_out,
                        hipStream_t st) {
  if (n_out <= 0) return 0;
  hipLaunchKernelGGL(gl_overlap_add_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, frames, wsq, wave,
                     T, n_fft, hop, n_out, 1.1754944e-38f);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
