// MCD evaluation kernels of config 5: batched dynamic time warping over anti-diagonals, RMS feature
// distance matrix, power spectrum / log glue of the MFCC front end (the STFT, mel and DCT steps are GEMMs).
//
// Reference call sites replaced: examples/s2s_trans/tasks/s2s_translation.py:414-460
// (batch_dynamic_time_warping: cumulative distance with first-minimum back pointers over
// [left, up-left, up], backtrace -> path map), :463-471 (compute_l2_dist / compute_rms_dist),
// :526-552 (batch_mel_cepstral_distortion's MFCC features; torchaudio in the reference).
#include "s2st_ops.h"
#include "s2st_prof.h"

namespace {

// one workgroup per batch element; the whole padded [M][N] matrix is processed like the reference does
__global__ __launch_bounds__(256) void dtw_kernel(const float* __restrict__ dist, const int* __restrict__ shapes,
                                                  float* __restrict__ cum, int* __restrict__ bp,
                                                  int* __restrict__ path, int M, int N) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* d = dist + (long)b * M * N;
  float* c = cum + (long)b * M * N;
  int* p = bp + (long)b * M * N;
  int* pm = path + (long)b * M * N;
  for (long i = tid; i < (long)M * N; i += 256) { p[i] = -1; pm[i] = 0; c[i] = 0.f; }
  __syncthreads();
  // first row / first column: sequential cumulative sums; torch.cumsum accumulates fp32 inputs in
  // double on the CPU (at::acc_type) and rounds each prefix to fp32 -- reproduced for bit-exactness
  if (tid == 0) {
    double a = 0.0;
    for (int j = 0; j < N; ++j) { a += (double)d[j]; c[j] = (float)a; p[j] = 0; }
  }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0;
    for (int i = 0; i < M; ++i) { a += (double)d[(long)i * N]; c[(long)i * N] = (float)a; p[(long)i * N] = 2; }
  }
  __syncthreads();
  for (int off = 2; off < M + N - 1; ++off) {
    // cells (i, j), i + j = off, 1 <= i < M, 1 <= j < N
    const int jlo = max(1, off - (M - 1)), jhi = min(N - 1, off - 1);
    for (int j = jlo + tid; j <= jhi; j += 256) {
      const int i = off - j;
      const float left = c[(long)i * N + j - 1], diag = c[(long)(i - 1) * N + j - 1], up = c[(long)(i - 1) * N + j];
      float v = left;
      int k = 0;
      if (diag < v) { v = diag; k = 1; }
      if (up < v) { v = up; k = 2; }
      p[(long)i * N + j] = k;
      c[(long)i * N + j] = v + d[(long)i * N + j];
    }
    __syncthreads();
  }
  if (tid == 0) {
    int i = shapes ? shapes[2 * b] - 1 : M - 1, j = shapes ? shapes[2 * b + 1] - 1 : N - 1;
    pm[(long)i * N + j] = 1;
    int steps = 1;
    while ((i != 0 || j != 0) && steps < 10000) {
      const int k = p[(long)i * N + j];
      if (k == 0) j -= 1;
      else if (k == 1) { i -= 1; j -= 1; }
      else i -= 1;
      if (i < 0 || j < 0) break;
      pm[(long)i * N + j] = 1;
      ++steps;
    }
  }
}

// out[i][j] = sqrt( sum_d (x1[i][d] - x2[j][d])^2 / D )
__global__ __launch_bounds__(256) void rms_dist_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                       float* __restrict__ out, int m, int n, int D, long ldo) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)m * n) return;
  const int i = (int)(t / n), j = (int)(t - (long)i * n);
  float a = 0.f;
  for (int k = 0; k < D; ++k) {
    const float e = x1[(long)i * D + k] - x2[(long)j * D + k];
    a += e * e;
  }
  out[(long)i * ldo + j] = sqrtf(a / D);
}

// P[t][f] = Y[t][f]^2 + Y[t][F + f]^2
__global__ __launch_bounds__(256) void power_spec_kernel(const float* __restrict__ Y, float* __restrict__ P, int T,
                                                         int F) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)T * F) return;
  const int t = (int)(i / F), f = (int)(i - (long)t * F);
  const float re = Y[(long)t * 2 * F + f], im = Y[(long)t * 2 * F + F + f];
  P[i] = re * re + im * im;
}

__global__ __launch_bounds__(256) void log_offset_kernel(float* __restrict__ x, long n, float eps) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = logf(x[i] + eps);
}

}  // namespace

int s2st_dtw(const float* dist, const int* shapes, int B, int M, int N, float* cum, int* backptr, int* pathmap,
             hipStream_t st) {
  if (B <= 0 || M <= 0 || N <= 0) return 0;
  S2ST_LAUNCH(dtw_kernel, dim3(B), dim3(256), 0, st, dist, shapes, cum, backptr, pathmap, M, N);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_rms_dist(const float* x1, const float* x2, float* out, int m, int n, int D, long ldo, hipStream_t st) {
  const long t = (long)m * n;
  if (t <= 0) return 0;
  S2ST_LAUNCH(rms_dist_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, st, x1, x2, out, m, n, D, ldo);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_power_spec(const float* Y, float* P, int T, int F, hipStream_t st) {
  const long t = (long)T * F;
  if (t <= 0) return 0;
  S2ST_LAUNCH(power_spec_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, st, Y, P, T, F);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_log_offset(float* x, long n, float eps, hipStream_t st) {
  if (n <= 0) return 0;
  S2ST_LAUNCH(log_offset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, eps);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
