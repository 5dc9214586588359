// Inference-only kernels of the AR mel decoder and the Griffin-Lim vocoder (config 5).
//
// Reference call sites replaced: fairseq/modules/multihead_attention.py:194-385 with
// incremental_state (one query per utterance against the cached keys / values, key padding mask,
// head-averaged weights of the alignment layer), fairseq/speech_generator_for_s2st.py:88-110
// (sigmoid of the stop logits, argmax alignment, global-CMVN de-normalisation) and
// fairseq/models/text_to_speech/vocoder.py:84-110 + fairseq/data/audio/audio_utils.py:259-271
// (Griffin-Lim: polar <-> rectangular spectra around the dense-DFT GEMMs, overlap-add).
#include "s2st_ops.h"
#include "s2st_prof.h"

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
// k_new / v_new (self-attention of a decoding step): the step's key / value rows [B][ld_new]; the workgroup first stores its
// (b, h) slice as row pos_new of the caches (no other workgroup reads or writes that slice), then attends over the cache
__global__ __launch_bounds__(256) void decode_attn_kernel(const float* __restrict__ q, long ldq,
                                                          float* __restrict__ kc, float* __restrict__ vc,
                                                          long ldk, long kbs, const int* __restrict__ klen, int nkeys,
                                                          int H, int dh, float scale, float* __restrict__ o, long ldo,
                                                          float* __restrict__ attn_mean, int S,
                                                          const float* __restrict__ k_new, const float* __restrict__ v_new,
                                                          long ld_new, int pos_new, const int* __restrict__ step_ptr) {
  if (step_ptr) { pos_new = *step_ptr; nkeys = pos_new + 1; }  // (replayable step: the step lives in device memory)
  __shared__ float p[DA_MAXS];
  __shared__ __attribute__((aligned(16))) float qs[256];
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) float part[16 * 256];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, tid = threadIdx.x;
  const int n = klen ? min((int)klen[b], nkeys) : nkeys;
  if (tid < dh) qs[tid] = q[(long)b * ldq + h * dh + tid] * scale;
  if (k_new && tid < dh) {
    kc[(long)b * kbs + (long)pos_new * ldk + h * dh + tid] = k_new[(long)b * ld_new + h * dh + tid];
    vc[(long)b * kbs + (long)pos_new * ldk + h * dh + tid] = v_new[(long)b * ld_new + h * dh + tid];
  }
  __syncthreads();  // (workgroup-scope: the stores above are visible to this workgroup's loads below)
  const float* kb = kc + (long)b * kbs + h * dh;
  const float* vb = vc + (long)b * kbs + h * dh;
  // 16 lanes per key, one float4 each: a 64-wide head row is one coalesced 256-byte read; 16 keys per pass
  const int g = tid >> 4, l4 = (tid & 15) * 4;
  // (4 keys per thread and pass, loads clamped instead of predicated so that they issue back to back:
  // the kernel is a chain of memory latencies otherwise)
  // (round 4: 8 keys per thread and pass -- 128 keys per workgroup and pass, every load of a pass issued before the first
  // multiply: the ~350 encoder positions of a cross-attention are 3 memory round trips instead of 6 x 2)
  constexpr int KP = 8;
  float mx = -INFINITY;
  if (dh <= 128 && dh % 64 == 0) {
    const int nd = dh / 64;  // 1 or 2 float4 chunks per lane
    for (int s0 = 0; s0 < n; s0 += 16 * KP) {
      float4 kv[2][KP];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < KP; ++i)
          kv[c][i] = *reinterpret_cast<const float4*>(kb + (long)min(s0 + g + 16 * i, n - 1) * ldk + min(l4 + 64 * c, dh - 4));
      float a[KP];
#pragma unroll
      for (int i = 0; i < KP; ++i) a[i] = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c >= nd) break;
        const float4 qv = *reinterpret_cast<const float4*>(qs + l4 + 64 * c);
#pragma unroll
        for (int i = 0; i < KP; ++i) a[i] += qv.x * kv[c][i].x + qv.y * kv[c][i].y + qv.z * kv[c][i].z + qv.w * kv[c][i].w;
      }
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        float v = a[i];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        const int s = s0 + g + 16 * i;
        if (s < n) {
          if (l4 == 0) p[s] = v;
          mx = fmaxf(mx, v);
        }
      }
    }
  } else
  for (int s0 = 0; s0 < n; s0 += 64) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = l4; d < dh; d += 64) {
      float4 kv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) kv[i] = *reinterpret_cast<const float4*>(kb + (long)min(s0 + g + 16 * i, n - 1) * ldk + d);
      const float4 qv = *reinterpret_cast<const float4*>(qs + d);
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] += qv.x * kv[i].x + qv.y * kv[i].y + qv.z * kv[i].z + qv.w * kv[i].w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = a[i];
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 1);
      const int s = s0 + g + 16 * i;
      if (s < n) {
        if (l4 == 0) p[s] = v;
        mx = fmaxf(mx, v);
      }
    }
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
  // o: 16 key slices x (16 lanes x float4) per 64 head columns, slices combined in a fixed order
  for (int d = l4; d < dh; d += 64) {
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < n; s0 += 16 * KP) {  // (8 value rows per thread in flight; same accumulation order as before)
      float4 vv[KP];
      float w[KP];
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        const int s = s0 + g + 16 * i;
        vv[i] = *reinterpret_cast<const float4*>(vb + (long)min(s, n - 1) * ldk + d);
        w[i] = s < n ? p[s] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        acc.x += w[i] * vv[i].x; acc.y += w[i] * vv[i].y; acc.z += w[i] * vv[i].z; acc.w += w[i] * vv[i].w;
      }
    }
    *reinterpret_cast<float4*>(part + g * dh + d) = acc;
  }
  __syncthreads();
  if (tid < dh) {
    float a = 0.f;
    for (int i = 0; i < 16; ++i) a += part[i * dh + tid];
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
__global__ __launch_bounds__(256) void exp_kernel(float* __restrict__ x, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = expf(x[i]);
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
// ---- batched Griffin-Lim on the bf16 matrix cores: U utterances with T_u = tl[u] <= Tmax frames, rows
// r = u * Tmax + t.  The dense-DFT GEMMs run as ONE bf16 GEMM each with the bf16x3 split folded into the
// contraction dimension: an fp32 row x (K values) is stored as the bf16 row [hi(x) | lo(x) | hi(x)] (3K) and
// the constant basis as [hi(b) | hi(b) | lo(b)], so the MFMA chain accumulates hi*hi + lo*hi + hi*lo in
// fp32 (~2^-17 relative, the accuracy of the fp32-operand "precise" GEMM) at the bf16 kernel's speed.
// Spectra are [rows][re(0..F) pad | im(0..F) pad] with the halves Fp = roundup(F, 16) apart; frames
// t >= T_u and pad columns are written as zeros.
__device__ __forceinline__ void store_split4(uint16_t* dst, long seg, const float v[4]) {
  uint2 hi, lo;
  split_bf16x4(v[0], v[1], v[2], v[3], hi, lo);
  *reinterpret_cast<uint2*>(dst) = hi;
  *reinterpret_cast<uint2*>(dst + seg) = lo;
  *reinterpret_cast<uint2*>(dst + 2 * seg) = hi;
}
// MODE 0: X = mag * exp(i ang) from the initial angles (aux = ang [rows][F]);
// MODE 1: X = mag * exp(i angle(Y)) from the re-analysed spectrum (aux = Y [rows][2 Fp])
template <int MODE>
__global__ __launch_bounds__(256) void gl_polar_split_kernel(const float* __restrict__ mag, const float* __restrict__ aux,
                                                             const int* __restrict__ tl, uint16_t* __restrict__ Xs,
                                                             int U, int F, int Fp, int Tmax) {
  const int q = Fp / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)U * Tmax * q) return;
  const int f4 = (int)(i % q) * 4;
  const long row = i / q;
  const int t = (int)(row % Tmax), u = (int)(row / Tmax);
  float re[4] = {0.f, 0.f, 0.f, 0.f}, im[4] = {0.f, 0.f, 0.f, 0.f};
  if (t < tl[u]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = f4 + e;
      if (f < F) {
        const float m = mag[row * F + f];
        const float a = MODE == 0 ? aux[row * F + f] : atan2f(aux[row * 2 * Fp + Fp + f], aux[row * 2 * Fp + f]);
        re[e] = m * cosf(a);
        im[e] = m * sinf(a);
      }
    }
  }
  uint16_t* xr = Xs + row * 6 * Fp;
  store_split4(xr + f4, 2 * Fp, re);
  store_split4(xr + Fp + f4, 2 * Fp, im);
}
// analysis frames of the reflect-padded waveforms, split for the STFT GEMM: As[row][3][n_fft]
__global__ __launch_bounds__(256) void gl_frame_split_kernel(const float* __restrict__ wave, const int* __restrict__ tl,
                                                             uint16_t* __restrict__ As, int U, int Tmax, int hop,
                                                             int n_fft, int Lw) {
  const int q = n_fft / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)U * Tmax * q) return;
  const int j4 = (int)(i % q) * 4;
  const long row = i / q;
  const int t = (int)(row % Tmax), u = (int)(row / Tmax);
  const int T = tl[u], n = hop * (T - 1), pad = n_fft / 2;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (t < T && n > pad) {
    const float* w = wave + (long)u * Lw;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int j = t * hop + j4 + e - pad;
      if (j < 0) j = -j;
      if (j >= n) j = 2 * (n - 1) - j;
      v[e] = w[j];
    }
  }
  store_split4(As + row * 3 * n_fft + j4, n_fft, v);
}
// per-utterance overlap-add of frames[u][t < T_u][n_fft]; wsq[T][..] is looked up per T_u through wsq_off
__global__ __launch_bounds__(256) void gl_overlap_add_b_kernel(const float* __restrict__ frames,
                                                               const float* __restrict__ wsq_all,
                                                               const long* __restrict__ wsq_off,
                                                               const int* __restrict__ tl, float* __restrict__ wave,
                                                               int U, int Tmax, int n_fft, int hop, int Lw, float tiny) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)U * Lw) return;
  const int u = (int)(i / Lw), q = (int)(i - (long)u * Lw);
  const int T = tl[u];
  float out = 0.f;
  if (q < hop * (T - 1)) {
    const int pos = q + n_fft / 2;
    int t1 = pos / hop;
    if (t1 > T - 1) t1 = T - 1;
    const int t0 = pos - n_fft + 1 <= 0 ? 0 : (pos - n_fft + 1 + hop - 1) / hop;
    float a = 0.f;
    for (int t = t0; t <= t1; ++t) a += frames[((long)u * Tmax + t) * n_fft + (pos - t * hop)];
    const float w = wsq_all[wsq_off[u] + pos];
    if (w > tiny) a /= w;
    out = a * ((float)n_fft / (float)hop);
  }
  wave[i] = out;
}

// The same attention for head widths 64 / 128 (round 4, second form).  The kernel above is a chain of memory round trips:
// key lengths and the query (-> LDS, barrier), key passes, two block-wide reductions, then value passes (x 2 column halves
// at width 128): 5.3 us for ONE key, 18 us for 315, and a batch's launch lasts as long as its longest utterance's
// workgroup (tools/decode_attn_bench.py, profiles/r04_t_decode_attn_bench.txt).  Here
//   * every 16-lane group keeps a RUNNING softmax over its own keys (s = g mod G): keys AND values of a pass are loaded
//     together, scores -> running maximum -> rescale -> accumulate; the G groups are merged once at the end through LDS in
//     group order (no block-wide reduction inside the loop, no atomics);
//   * the first pass's loads wait for the utterance's key count only: the query comes straight from global memory, and the
//     step's own key / value row is taken from k_new / v_new in registers (and stored to the cache for the later steps)
//     instead of being stored, fenced by a barrier and read back.  (Clamping the rows to the cache's extent instead of
//     the utterance's length, so that not even the key count is waited for, was tried: every workgroup then fetches a full
//     pass of distinct rows -- 67 MB per launch instead of 23 MB for the bench batch -- and the launch takes 20 us);
//   * NT = 1024 threads (64 groups: 256 keys per pass in fp32, 512 with bf16 rows) and KT = bf16raw (the static
//     cross-attention rows as bf16) are built and tested but NOT the default: measured on the bench batch (64 utterances x 4
//     heads of width 128, 8 - 315 keys, tools/decode_attn_bench.py) 16.3 us and 13.9 us against 13.3 us for 256 threads on
//     fp32 rows -- a launch lasts as long as its longest utterance's workgroup, which pulls its rows at ~20 B/clk whatever
//     their type, and the wider forms start later (5.8 us for one key with 256 threads on fp32, 8.4 with bf16, 9.3 with
//     1024 threads).
template <typename KT>
struct RawRow;
template <>
struct RawRow<float> {
  typedef float4 T;
  static __device__ __forceinline__ float4 cvt(const float4& r) { return r; }
};
template <>
struct RawRow<bf16raw> {
  typedef uint2 T;
  static __device__ __forceinline__ float4 cvt(const uint2& u) {
    return float4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                  __uint_as_float(u.y & 0xffff0000u)};
  }
};

template <typename KT, int ND, int NT>
__global__ __launch_bounds__(NT) void decode_attn_fast_kernel(const float* __restrict__ q, long ldq, KT* __restrict__ kc,
                                                              KT* __restrict__ vc, long ldk, long kbs,
                                                              const int* __restrict__ klen, int nkeys, int H, float scale,
                                                              float* __restrict__ o, long ldo, float* __restrict__ attn_mean,
                                                              int S, const float* __restrict__ k_new,
                                                              const float* __restrict__ v_new, long ld_new, int pos_new,
                                                              const int* __restrict__ step_ptr) {
  if (step_ptr) { pos_new = *step_ptr; nkeys = pos_new + 1; }  // (replayable step: the step lives in device memory)
  constexpr int dh = 64 * ND, G = NT / 16;
  constexpr int KP = (sizeof(KT) == 2 ? 16 : 8) / (NT / 256 > 2 ? 2 : 1);
  typedef typename RawRow<KT>::T Raw;
  __shared__ float p[DA_MAXS];
  __shared__ float gm[G], gl[G];
  __shared__ __attribute__((aligned(16))) float part[G * dh];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, tid = threadIdx.x;
  const int g = tid >> 4, l4 = (tid & 15) * 4;
  const KT* kb = kc + (long)b * kbs + h * dh;
  const KT* vb = vc + (long)b * kbs + h * dh;
  const int row_new = (sizeof(KT) == 4 && k_new) ? pos_new : -1;
  // ---- everything below up to the first use is independent loads ----
  const int n = klen ? min((int)klen[b], nkeys) : nkeys;
  float4 qv[ND], kn[ND], vn[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) {
    qv[c] = *reinterpret_cast<const float4*>(q + (long)b * ldq + h * dh + l4 + 64 * c);
    if (row_new >= 0) {
      kn[c] = *reinterpret_cast<const float4*>(k_new + (long)b * ld_new + h * dh + l4 + 64 * c);
      vn[c] = *reinterpret_cast<const float4*>(v_new + (long)b * ld_new + h * dh + l4 + 64 * c);
    } else {
      kn[c] = vn[c] = float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  float m_run = -INFINITY, l_run = 0.f;
  float4 acc[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) acc[c] = float4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < n; s0 += G * KP) {
    Raw kr[ND][KP], vr[ND][KP];
#pragma unroll
    for (int c = 0; c < ND; ++c)
#pragma unroll
      for (int i = 0; i < KP; ++i)
        kr[c][i] = *reinterpret_cast<const Raw*>(kb + (long)min(s0 + g + G * i, n - 1) * ldk + l4 + 64 * c);
#pragma unroll
    for (int c = 0; c < ND; ++c)
#pragma unroll
      for (int i = 0; i < KP; ++i)
        vr[c][i] = *reinterpret_cast<const Raw*>(vb + (long)min(s0 + g + G * i, n - 1) * ldk + l4 + 64 * c);
    if (s0 == 0) {
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        qv[c].x *= scale; qv[c].y *= scale; qv[c].z *= scale; qv[c].w *= scale;
      }
    }
    float a[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) a[i] = 0.f;
#pragma unroll
    for (int c = 0; c < ND; ++c)
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        float4 k4 = RawRow<KT>::cvt(kr[c][i]);
        if (s0 + g + G * i == row_new) k4 = kn[c];
        a[i] += qv[c].x * k4.x + qv[c].y * k4.y + qv[c].z * k4.z + qv[c].w * k4.w;
      }
    float pm = -INFINITY;
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      float v = a[i];
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 1);
      const int s = s0 + g + G * i;
      if (s < n) {
        if (attn_mean && l4 == 0) p[s] = v;
        pm = fmaxf(pm, v);
      } else {
        v = -INFINITY;
      }
      a[i] = v;
    }
    if (pm == -INFINITY) continue;  // (none of this group's keys of the pass exists)
    const float m_new = fmaxf(m_run, pm);
    const float sc = __expf(m_run - m_new);  // (first pass: exp(-inf) = 0)
    l_run *= sc;
#pragma unroll
    for (int c = 0; c < ND; ++c) {
      acc[c].x *= sc; acc[c].y *= sc; acc[c].z *= sc; acc[c].w *= sc;
    }
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const bool on = a[i] != -INFINITY;
      const float e = on ? __expf(a[i] - m_new) : 0.f;
      l_run += e;
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        float4 v4 = RawRow<KT>::cvt(vr[c][i]);
        if (s0 + g + G * i == row_new) v4 = vn[c];
        if (!on) v4 = float4{0.f, 0.f, 0.f, 0.f};  // (a row behind the utterance's end may hold anything)
        acc[c].x += e * v4.x; acc[c].y += e * v4.y; acc[c].z += e * v4.z; acc[c].w += e * v4.w;
      }
    }
    m_run = m_new;
  }
  if constexpr (sizeof(KT) == 4) {
    if (row_new >= 0 && g == 0) {  // this step's rows into the caches (read by the later steps' launches)
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        *reinterpret_cast<float4*>(kc + (long)b * kbs + (long)row_new * ldk + h * dh + l4 + 64 * c) = kn[c];
        *reinterpret_cast<float4*>(vc + (long)b * kbs + (long)row_new * ldk + h * dh + l4 + 64 * c) = vn[c];
      }
    }
  }
  if (l4 == 0) {
    gm[g] = m_run;
    gl[g] = l_run;
  }
#pragma unroll
  for (int c = 0; c < ND; ++c) *reinterpret_cast<float4*>(part + g * dh + l4 + 64 * c) = acc[c];
  __syncthreads();
  // merge of the G groups, in group order
  float M = gm[0];
  for (int i = 1; i < G; ++i) M = fmaxf(M, gm[i]);
  float Lsum = 0.f;
  for (int i = 0; i < G; ++i) Lsum += gm[i] == -INFINITY ? 0.f : gl[i] * __expf(gm[i] - M);
  const float inv = Lsum > 0.f ? 1.f / Lsum : 0.f;
  if (tid < dh) {
    float a = 0.f;
    for (int i = 0; i < G; ++i) a += gm[i] == -INFINITY ? 0.f : part[i * dh + tid] * __expf(gm[i] - M);
    o[(long)b * ldo + h * dh + tid] = a * inv;
  }
  if (attn_mean)
    for (int s = tid; s < n; s += NT) atomicAdd(attn_mean + (long)b * S + s, __expf(p[s] - M) * inv / H);
}

// ------------------------------------------------------------------------------------------------
// Griffin-Lim with FFTs (round 4).  The reference's STFT / inverse STFT are dense Fourier-basis convolutions
// (audio_utils.py:226-271: basis = [Re; Im] of fft(eye(n_fft)) * window; vocoder.py:56-98: pinverse(n_fft / hop * basis)):
//   * the analysis is exactly rfft(window * frame);
//   * the synthesis basis pinverse(s B)^T equals the plain inverse real FFT / s: B^T B = (N/2) I + E with E[n][m] = 1 for
//     n - m even, whose inverse turns B^T X = Re sum_{k <= N/2} X_k W^{kn} into (1/N) [X_0 + X_{N/2} (-1)^n + 2 sum_{0<k<N/2}
//     Re X_k W^{kn}] (imaginary parts of the DC and Nyquist bins dropped) -- checked numerically against numpy's pinv to 1e-16.
// So both directions are N-point real FFTs: O(N log N) per frame instead of the 2 N (N + 2) multiply-adds of the dense
// contraction (x 3 in the bf16x3 GEMM form that rounds 1 - 3 used: 137 ms of the 237 ms one 16-utterance batch took).
// One workgroup transforms TWO frames at once as the real and imaginary part of one complex N-point FFT (Stockham
// autosort, radix 4 with a final radix 2 when log2 N is odd, in LDS: 8 N bytes + the twiddle table) and separates /
// merges the two spectra through the Hermitian symmetry.  N = 256 ... 2048 (powers of two); other n_fft keep the GEMM path.
// ------------------------------------------------------------------------------------------------
struct cplx { float x, y; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return cplx{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return cplx{a.x - b.x, a.y - b.y}; }

// LDS arrays are PADDED: logical element i lives at P(i) = i + (i >> 5).  The Stockham passes scatter their results with
// power-of-two strides (4, 16, 64 elements of 8 bytes) and read twiddles at strides of N / (4 Ns): without the pad 8 - 16
// lanes of a wave hit one bank (the first measurement of these kernels on MI355X: 390 us per 44.5 k-frame STFT, ~10 k cycles
// of a CU per frame pair against ~2 k of LDS traffic).
__device__ __forceinline__ int fpad(int i) { return i + (i >> 5); }
template <int N>
struct FftLds {
  static constexpr int SIZE = N + N / 32 + 1;
};

// in-place complex FFT of the padded LDS array buf (logical 0 .. N), 256 threads; tw (padded) logical j = exp(-2 pi i j / N);
// INV: conjugate transform (unscaled).  Stockham autosort passes of radix 8 while a factor 8 is left, then one radix-4 or
// radix-2 pass (2048 = 8 * 8 * 8 * 4: FOUR LDS round trips; the first form of this kernel ran five radix-4 passes and a
// radix-2 one with run-time strides).  Pass with Ns done: butterfly j (0 .. N / R) reads x_r = buf[j + r N / R] * w^(r k)
// with k = j mod Ns, w = exp(-+2 pi i / (R Ns)), and writes its R outputs to (j - k) R + k + s Ns.
template <bool INV>
__device__ __forceinline__ cplx mul_mi(cplx d) { return INV ? cplx{-d.y, d.x} : cplx{d.y, -d.x}; }  // d * (-+ i)
template <bool INV>
__device__ __forceinline__ cplx mul_w8(cplx d) {  // d * exp(-+ i pi / 4)
  constexpr float h = 0.70710678118654752440f;
  return INV ? cplx{h * (d.x - d.y), h * (d.x + d.y)} : cplx{h * (d.x + d.y), h * (d.y - d.x)};
}
template <bool INV>
__device__ __forceinline__ cplx mul_w83(cplx d) {  // d * exp(-+ 3 i pi / 4)
  constexpr float h = 0.70710678118654752440f;
  return INV ? cplx{-h * (d.x + d.y), h * (d.x - d.y)} : cplx{h * (d.y - d.x), -h * (d.x + d.y)};
}
template <int R, bool INV>
__device__ __forceinline__ void dft_small(cplx* v) {
  if constexpr (R == 2) {
    const cplx a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  } else if constexpr (R == 4) {
    const cplx e0 = cadd(v[0], v[2]), e1 = csub(v[0], v[2]), e2 = cadd(v[1], v[3]), e3 = mul_mi<INV>(csub(v[1], v[3]));
    v[0] = cadd(e0, e2);
    v[1] = cadd(e1, e3);
    v[2] = csub(e0, e2);
    v[3] = csub(e1, e3);
  } else {
    // even outputs: 4-point transform of x_n + x_{n+4}; odd outputs: of (x_n - x_{n+4}) * w8^n
    cplx c[4] = {cadd(v[0], v[4]), cadd(v[1], v[5]), cadd(v[2], v[6]), cadd(v[3], v[7])};
    cplx d[4] = {csub(v[0], v[4]), mul_w8<INV>(csub(v[1], v[5])), mul_mi<INV>(csub(v[2], v[6])), mul_w83<INV>(csub(v[3], v[7]))};
    dft_small<4, INV>(c);
    dft_small<4, INV>(d);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[2 * q] = c[q];
      v[2 * q + 1] = d[q];
    }
  }
}
template <int N, int NS, int R, bool INV>
__device__ __forceinline__ void fft_pass(cplx* buf, const cplx* tw, int tid) {
  constexpr int NB = N / R, PER = (NB + 255) / 256, TS = N / (R * NS);
  cplx v[PER][R];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int j = tid + 256 * i;
    if (NB % 256 == 0 || j < NB) {
      const int k = j & (NS - 1);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        cplx x = buf[fpad(j + r * NB)];
        if (NS > 1 && r > 0) {
          cplx w = tw[fpad(r * k * TS)];
          if (INV) w.y = -w.y;
          x = cmul(x, w);
        }
        v[i][r] = x;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int j = tid + 256 * i;
    if (NB % 256 == 0 || j < NB) {
      const int k = j & (NS - 1);
      dft_small<R, INV>(v[i]);
      const int j0 = (j - k) * R + k;
#pragma unroll
      for (int r = 0; r < R; ++r) buf[fpad(j0 + r * NS)] = v[i][r];
    }
  }
  __syncthreads();
}
template <int N, int NS, bool INV>
__device__ __forceinline__ void fft_passes(cplx* buf, const cplx* tw, int tid) {
  if constexpr (NS < N) {
    constexpr int R = N / NS >= 8 ? 8 : N / NS;
    fft_pass<N, NS, R, INV>(buf, tw, tid);
    fft_passes<N, NS * R, INV>(buf, tw, tid);
  }
}
template <int N, bool INV>
__device__ void fft_lds(cplx* buf, const cplx* tw, int tid) {
  fft_passes<N, 1, INV>(buf, tw, tid);
}

// STFT of the reflect-padded waveforms + phase projection, per PAIR of frames (rows m0 = 2 * pair, m0 + 1 of the
// flattened [U][Tmax] frame index): X[m][k] = mag[m][k] * Y_k / |Y_k| with Y = rfft(window * frame) -- the reference's
// mag * (cos, sin)(atan2(Im Y, Re Y)) (audio_utils.py:259-271, vocoder.py:104-107; |Y| = 0: angle 0).  Rows t >= T_u: zeros.
// A workgroup walks several pairs; the NEXT pair's samples and this pair's magnitudes are fetched into registers before
// the transform starts, so the memory round trips run under the butterflies (first version: load -> transform -> load ->
// store in sequence, ~18 us per pair and workgroup with four workgroups per CU).
template <int N>
struct PairInfo {
  int tt[2], uu[2], TT[2];
  bool on[2];
  __device__ __forceinline__ void set(long pair, long M, int Tmax, const int* tl, int hop, bool need_len) {
    for (int h = 0; h < 2; ++h) {
      const long m = 2 * pair + h;
      uu[h] = m < M ? (int)(m / Tmax) : 0;
      tt[h] = m < M ? (int)(m - (long)uu[h] * Tmax) : 0;
      TT[h] = tl[uu[h]];
      on[h] = m < M && tt[h] < TT[h] && (!need_len || hop * (TT[h] - 1) > N / 2);
    }
  }
};

template <int N>
__global__ __launch_bounds__(256) void gl_stft_project_kernel(const float* __restrict__ wave, const int* __restrict__ tl,
                                                              const float* __restrict__ win, const cplx* __restrict__ twg,
                                                              const float* __restrict__ mag, cplx* __restrict__ X, int U,
                                                              int Tmax, int hop, int Lw, long npairs) {
  __shared__ cplx buf[FftLds<N>::SIZE];
  __shared__ cplx tw[FftLds<N>::SIZE];
  const int tid = threadIdx.x;
  constexpr int F = N / 2 + 1, PN = N / 256, PF = (F + 255) / 256;
  for (int j = tid; j < N; j += 256) tw[fpad(j)] = twg[j];
  const long M = (long)U * Tmax;
  float wn[PN];
#pragma unroll
  for (int i = 0; i < PN; ++i) wn[i] = win[tid + 256 * i];
  cplx xr[PN];
  auto fetch = [&](const PairInfo<N>& pi) {
#pragma unroll
    for (int i = 0; i < PN; ++i) {
      const int n = tid + 256 * i;
      float v[2] = {0.f, 0.f};
      for (int h = 0; h < 2; ++h) {
        if (!pi.on[h]) continue;
        const int len = hop * (pi.TT[h] - 1);
        int j = pi.tt[h] * hop + n - N / 2;
        if (j < 0) j = -j;
        if (j >= len) j = 2 * (len - 1) - j;
        v[h] = wave[(long)pi.uu[h] * Lw + j];
      }
      xr[i] = cplx{v[0] * wn[i], v[1] * wn[i]};
    }
  };
  PairInfo<N> cur, nxt;
  long pair = blockIdx.x;
  if (pair < npairs) {
    cur.set(pair, M, Tmax, tl, hop, true);
    fetch(cur);
  }
  for (; pair < npairs; pair += gridDim.x) {
    const long m0 = 2 * pair;
    __syncthreads();  // (the previous pair's readers are done with buf)
#pragma unroll
    for (int i = 0; i < PN; ++i) buf[fpad(tid + 256 * i)] = xr[i];
    // this pair's magnitudes and the next pair's samples: in flight during the transform
    float mg[PF][2];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int k = tid + 256 * i;
      for (int h = 0; h < 2; ++h) mg[i][h] = (k < F && cur.on[h]) ? mag[(m0 + h) * F + k] : 0.f;
    }
    const long np_ = pair + gridDim.x;
    if (np_ < npairs) {
      nxt.set(np_, M, Tmax, tl, hop, true);
      fetch(nxt);
    }
    __syncthreads();
    fft_lds<N, false>(buf, tw, tid);
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int k = tid + 256 * i;
      if (k >= F) continue;
      const cplx zk = buf[fpad(k)], zn = buf[fpad((N - k) & (N - 1))];
      // Y1 = (Z_k + conj Z_{N-k}) / 2 ; Y2 = (Z_k - conj Z_{N-k}) / (2 i)
      const cplx y[2] = {cplx{0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)}, cplx{0.5f * (zk.y + zn.y), 0.5f * (zn.x - zk.x)}};
      for (int h = 0; h < 2; ++h) {
        const long m = m0 + h;
        if (m >= M) continue;
        cplx o{0.f, 0.f};
        if (cur.on[h]) {
          const float a2 = y[h].x * y[h].x + y[h].y * y[h].y;
          if (a2 > 0.f) {
            const float r = mg[i][h] * rsqrtf(a2);
            o = cplx{y[h].x * r, y[h].y * r};
          } else {
            o = cplx{mg[i][h], 0.f};
          }
        }
        X[m * F + k] = o;
      }
    }
    cur = nxt;
  }
}

// inverse: frames[m][n] = window[n] * (hop / N) * irfft(X[m])[n] for the two rows of a pair (rows t >= T_u: zeros);
// the overlap-add kernel above turns the frames into the waveforms.  The next pair's spectra are fetched ahead likewise.
template <int N>
__global__ __launch_bounds__(256) void gl_istft_frames_kernel(const cplx* __restrict__ X, const int* __restrict__ tl,
                                                              const float* __restrict__ win, const cplx* __restrict__ twg,
                                                              float* __restrict__ frames, int U, int Tmax, int hop, long npairs) {
  __shared__ cplx buf[FftLds<N>::SIZE];
  __shared__ cplx tw[FftLds<N>::SIZE];
  const int tid = threadIdx.x;
  constexpr int F = N / 2 + 1, PN = N / 256, PF = (F + 255) / 256;
  for (int j = tid; j < N; j += 256) tw[fpad(j)] = twg[j];
  const float sc = (float)hop / ((float)N * (float)N);
  const long M = (long)U * Tmax;
  float wn[PN];
#pragma unroll
  for (int i = 0; i < PN; ++i) wn[i] = win[tid + 256 * i] * sc;
  cplx xa[PF], xb[PF];
  auto fetch = [&](const PairInfo<N>& pi, long m0) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int k = tid + 256 * i;
      xa[i] = (k < F && pi.on[0]) ? X[m0 * F + k] : cplx{0.f, 0.f};
      xb[i] = (k < F && pi.on[1]) ? X[(m0 + 1) * F + k] : cplx{0.f, 0.f};
    }
  };
  PairInfo<N> cur, nxt;
  long pair = blockIdx.x;
  if (pair < npairs) {
    cur.set(pair, M, Tmax, tl, hop, false);
    fetch(cur, 2 * pair);
  }
  for (; pair < npairs; pair += gridDim.x) {
    const long m0 = 2 * pair;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int k = tid + 256 * i;
      if (k >= F) continue;
      cplx a = xa[i], b = xb[i];
      if (k == 0 || k == N / 2) a.y = b.y = 0.f;  // (the basis' sine rows of DC / Nyquist are zero)
      buf[fpad(k)] = cplx{a.x - b.y, a.y + b.x};                             // Z_k     = X1_k + i X2_k
      if (k > 0 && k < N / 2) buf[fpad(N - k)] = cplx{a.x + b.y, b.x - a.y};  // Z_{N-k} = conj X1_k + i conj X2_k
    }
    const long np_ = pair + gridDim.x;
    if (np_ < npairs) {
      nxt.set(np_, M, Tmax, tl, hop, false);
      fetch(nxt, 2 * np_);
    }
    __syncthreads();
    fft_lds<N, true>(buf, tw, tid);
#pragma unroll
    for (int i = 0; i < PN; ++i) {
      const int n = tid + 256 * i;
      const cplx z = buf[fpad(n)];
      if (m0 < M) frames[m0 * N + n] = cur.on[0] ? z.x * wn[i] : 0.f;
      if (m0 + 1 < M) frames[(m0 + 1) * N + n] = cur.on[1] ? z.y * wn[i] : 0.f;
    }
    cur = nxt;
  }
}

// inverse transform AND overlap-add in one launch (the 367 MB of synthesis frames per iteration -- written by the kernel
// above, read back by gl_overlap_add_b -- never exist).  Workgroup (b, u) owns the output samples q in [b S, (b + 1) S) of
// utterance u (S = R hops) and inverse-transforms every frame that touches them in ascending order: its own R frames plus
// the ceil(N / hop) in front whose tails reach in (recomputed by the neighbouring workgroup too: +11 % transforms at R = 64
// for n_fft 2048 / hop 300).  The accumulator is a CIRCULAR LDS array of C >= N + 2 hop floats (a power of two): once a pair
// (t, t + 1) is added, the samples in front of (t + 2) hop are final -- they are normalised, written out and their slots
// cleared for the samples C further on.  (First form: an S-float accumulator, 72 KB of LDS per workgroup, two workgroups per
// CU: 329 us per launch against 136 + 164 for the two-kernel form; the transforms are latency-bound and want 3 - 4
// workgroups per CU.)  Frames are added in ascending order with a barrier between the two frames of a pair, so every sample
// sees the additions the stand-alone overlap-add makes, in its order (which two frames share a complex transform differs
// from the kernel above, so a frame's last bits can: 1e-8 relative, tests/test_inference.py).
template <int N>
__global__ __launch_bounds__(256) void gl_istft_ola_kernel(const cplx* __restrict__ X, const int* __restrict__ tl,
                                                           const float* __restrict__ win, const cplx* __restrict__ twg,
                                                           const float* __restrict__ wsq_all, const long* __restrict__ wsq_off,
                                                           float* __restrict__ wave, int Tmax, int hop, int Lw, int S, int C,
                                                           float tiny) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  cplx* buf = reinterpret_cast<cplx*>(smem);
  cplx* tw = buf + FftLds<N>::SIZE;
  float* acc = reinterpret_cast<float*>(tw + FftLds<N>::SIZE);
  const int tid = threadIdx.x, u = blockIdx.y;
  constexpr int F = N / 2 + 1, PN = N / 256, PF = (F + 255) / 256;
  const int T = tl[u];
  const int q0 = blockIdx.x * S, q1 = min(q0 + S, Lw);
  const int len = hop * (T - 1);  // samples of this utterance (vocoder.py:95-97: the n_fft / 2 borders are trimmed)
  float* out = wave + (long)u * Lw;
  if (q0 >= len) {
    for (int q = q0 + tid; q < q1; q += 256) out[q] = 0.f;
    return;
  }
  for (int q = max(q0, len) + tid; q < q1; q += 256) out[q] = 0.f;  // (behind the utterance's end)
  for (int j = tid; j < N; j += 256) tw[fpad(j)] = twg[j];
  for (int i = tid; i < C; i += 256) acc[i] = 0.f;
  const int cm = C - 1;
  const float sc = (float)hop / ((float)N * (float)N);
  float wn[PN];
#pragma unroll
  for (int i = 0; i < PN; ++i) wn[i] = win[tid + 256 * i] * sc;
  const int P0 = q0 + N / 2, P1 = min(q1, len) + N / 2;  // owned positions of the untrimmed signal
  const int t_lo = P0 - N + 1 <= 0 ? 0 : (P0 - N + hop) / hop;
  const int t_hi = min(T - 1, (P1 - 1) / hop);
  const cplx* Xu = X + (long)u * Tmax * F;
  const float* wsq = wsq_all + wsq_off[u];
  const float gain = (float)N / (float)hop;
  cplx xa[PF], xb[PF];
  auto fetch = [&](int t) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int k = tid + 256 * i;
      xa[i] = (k < F && t <= t_hi) ? Xu[(long)t * F + k] : cplx{0.f, 0.f};
      xb[i] = (k < F && t + 1 <= t_hi) ? Xu[(long)(t + 1) * F + k] : cplx{0.f, 0.f};
    }
  };
  fetch(t_lo);
  int done = t_lo * hop;  // positions in front of `done` are final and flushed
  for (int t = t_lo; t <= t_hi; t += 2) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int k = tid + 256 * i;
      if (k >= F) continue;
      cplx a = xa[i], b = xb[i];
      if (k == 0 || k == N / 2) a.y = b.y = 0.f;
      buf[fpad(k)] = cplx{a.x - b.y, a.y + b.x};
      if (k > 0 && k < N / 2) buf[fpad(N - k)] = cplx{a.x + b.y, b.x - a.y};
    }
    if (t + 2 <= t_hi) fetch(t + 2);
    __syncthreads();
    fft_lds<N, true>(buf, tw, tid);
    float zx[PN], zy[PN];
    {
#pragma clang fp contract(off)  // (product rounded, then added: what the frames-through-HBM form does; no fused multiply-add)
#pragma unroll
      for (int i = 0; i < PN; ++i) {
        const cplx z = buf[fpad(tid + 256 * i)];
        zx[i] = z.x * wn[i];
        zy[i] = z.y * wn[i];
      }
#pragma unroll
      for (int i = 0; i < PN; ++i) {
        const int sl = (t * hop + tid + 256 * i) & cm;
        acc[sl] = acc[sl] + zx[i];
      }
    }
    __syncthreads();
    if (t + 1 <= t_hi) {
#pragma unroll
      for (int i = 0; i < PN; ++i) {
        const int sl = ((t + 1) * hop + tid + 256 * i) & cm;
        acc[sl] = acc[sl] + zy[i];
      }
    }
    __syncthreads();
    // final now: everything in front of the next pair's first position (after the last pair: up to the block's end)
    const int upto = t + 2 <= t_hi ? (t + 2) * hop : P1;
    for (int p = done + tid; p < upto; p += 256) {
      const int sl = p & cm;
      if (p >= P0 && p < P1) {
        float a = acc[sl];
        const float w = wsq[p];
        if (w > tiny) a /= w;
        out[p - N / 2] = a * gain;
      }
      acc[sl] = 0.f;
    }
    done = upto;
  }
}

// The initial phases on the device.  The reference draws np.random.rand(F, T_u) per utterance from numpy's global generator and
// takes np.angle(np.exp(2j pi u)) (vocoder.py:101-102) = 2 pi u wrapped into (-pi, pi], in double, then casts to the
// spectrogram's dtype.  `uni`: those uniform draws, as drawn (utterance u's [F][T_u] block at uni + uoff[u]): the wrap, the
// cast, the transposition to frame-major and mag * (cos, sin) happen here (the host only runs the generator -- rounds 1 - 3
// also wrapped, cast and transposed 45 M numbers per 64 utterances there: 2/3 of the vocoder's wall time).  uni == nullptr:
// the draws themselves come from the counter-based hash of (seed, utterance, bin, frame) -- same distribution, not numpy's
// stream (GriffinLim(phase_rng="device")).  One thread per (u, k, t), t fastest: coalesced reads of uni.
__global__ __launch_bounds__(256) void gl_polar_u_kernel(const float* __restrict__ mag, const double* __restrict__ uni,
                                                         const long* __restrict__ uoff, const int* __restrict__ tl,
                                                         uint64_t seed, cplx* __restrict__ X, int U, int F, int Tmax) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)U * F * Tmax) return;
  const int t = (int)(i % Tmax);
  const long r = i / Tmax;
  const int k = (int)(r % F), u = (int)(r / F);
  const int T = tl[u];
  cplx o{0.f, 0.f};
  if (t < T) {
    double un;
    if (uni) un = uni[uoff[u] + (long)k * T + t];
    else un = (double)(mix32(seed + (uint64_t)u, (uint64_t)k * (uint64_t)Tmax + (uint64_t)t) >> 8) * (1.0 / 16777216.0);
    double a = 6.283185307179586476925286766559 * un;
    if (a > 3.141592653589793238462643383279) a -= 6.283185307179586476925286766559;
    const float af = (float)a, m = mag[((long)u * Tmax + t) * F + k];
    o = cplx{m * cosf(af), m * sinf(af)};
  }
  X[((long)u * Tmax + t) * F + k] = o;
}

// X[m][k] = mag[m][k] * (cos, sin)(ang[m][k]) (the initial phases, vocoder.py:101-103); rows t >= T_u: zeros
__global__ __launch_bounds__(256) void gl_polar_c_kernel(const float* __restrict__ mag, const float* __restrict__ ang,
                                                         const int* __restrict__ tl, cplx* __restrict__ X, int U, int F, int Tmax) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)U * Tmax * F) return;
  const long row = i / F;
  const int t = (int)(row % Tmax), u = (int)(row / Tmax);
  cplx o{0.f, 0.f};
  if (t < tl[u]) {
    const float m = mag[i], a = ang[i];
    o = cplx{m * cosf(a), m * sinf(a)};
  }
  X[i] = o;
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

// fairseq's reorder_incremental_state for the cached keys / values of a beam search (sequence_generator.py:393-400,
// multihead_attention.py:387-403): dst[n][b][0 : valid) = src[n][idx[b]][0 : valid) for the nb (layer, K | V) arrays.
namespace {
__global__ __launch_bounds__(256) void cache_reorder_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                            const int* __restrict__ idx, int Bb, long row4, long valid4) {
  const int b = blockIdx.x, n = blockIdx.y;
  const float4* s = src + ((long)n * Bb + idx[b]) * row4;
  float4* d = dst + ((long)n * Bb + b) * row4;
  for (long i = threadIdx.x; i < valid4; i += 256) d[i] = s[i];
}
}  // namespace
int s2st_cache_reorder(const float* src, float* dst, const int* idx, int nb, int Bb, long row_floats, long valid_floats,
                       hipStream_t st) {
  if (nb <= 0 || Bb <= 0 || valid_floats <= 0) return 0;
  if (row_floats % 4 || valid_floats % 4 || ((uintptr_t)src % 16) || ((uintptr_t)dst % 16) || !idx) return S2ST_ERR_ARG;
  S2ST_LAUNCH(cache_reorder_kernel, dim3(Bb, nb), dim3(256), 0, st, reinterpret_cast<const float4*>(src),
              reinterpret_cast<float4*>(dst), idx, Bb, row_floats / 4, valid_floats / 4);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_decode_attn(const float* q, long ldq, float* kc, float* vc, long ldk, long kbs, const int* klen,
                     int nkeys, int B, int H, int dh, float scale, float* o, long ldo, float* attn_mean, int S,
                     hipStream_t st, const float* k_new, const float* v_new, long ld_new, int pos_new, int kv_bf16,
                     const int* step_ptr) {
  if (B <= 0) return 0;
  if (nkeys > DA_MAXS || dh > 256 || dh % 4 || 256 % dh) return S2ST_ERR_SHAPE;
  if ((k_new != nullptr) != (v_new != nullptr) || (k_new && !step_ptr && (pos_new < 0 || pos_new >= nkeys))) return S2ST_ERR_ARG;
  if (kv_bf16 && (k_new || (dh != 64 && dh != 128) || ldk % 4)) return S2ST_ERR_ARG;  // (bf16 caches: static rows only)
  if (nkeys < 1) return S2ST_ERR_ARG;
  if (attn_mean) hipMemsetAsync(attn_mean, 0, sizeof(float) * (size_t)B * S, st);
  // (round 4's A/B forms are closed and gone: the two-pass first kernel stays only as the fallback for head widths other
  //  than 64 / 128; 64 key groups per workgroup measured 16.3 against 13.3 us on the bench batch, profiles/r04_t_decode_attn_bench.txt)
  const bool fast_ok = (dh == 64 || dh == 128) && ldk % 4 == 0 && ldq % 4 == 0 && (!k_new || ld_new % 4 == 0);
  if (kv_bf16 && !fast_ok) return S2ST_ERR_ARG;
#define S2ST_DA_LAUNCH(KT_, ND_, NT_, kp, vp)                                                                               \
  S2ST_LAUNCH((decode_attn_fast_kernel<KT_, ND_, NT_>), dim3(B * H), dim3(NT_), 0, st, q, ldq, kp, vp, ldk, kbs, klen, nkeys, H, \
              scale, o, ldo, attn_mean, S, k_new, v_new, ld_new, pos_new, step_ptr)
  if (!fast_ok) {
    S2ST_LAUNCH(decode_attn_kernel, dim3(B * H), dim3(256), 0, st, q, ldq, kc, vc, ldk, kbs, klen, nkeys, H, dh,
                scale, o, ldo, attn_mean, S, k_new, v_new, ld_new, pos_new, step_ptr);
  } else if (kv_bf16) {
    bf16raw* kh = reinterpret_cast<bf16raw*>(kc);
    bf16raw* vh = reinterpret_cast<bf16raw*>(vc);
    if (dh == 64) S2ST_DA_LAUNCH(bf16raw, 1, 256, kh, vh);
    else S2ST_DA_LAUNCH(bf16raw, 2, 256, kh, vh);
  } else {
    if (dh == 64) S2ST_DA_LAUNCH(float, 1, 256, kc, vc);
    else S2ST_DA_LAUNCH(float, 2, 256, kc, vc);
  }
#undef S2ST_DA_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

namespace {
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                         float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = a[0] * x[i];
}
}  // namespace

// y = a[0] * x (a: one device scalar): the alpha-scaled position table of a decoding run
int s2st_scale_rows(const float* x, const float* a, float* y, long n, hipStream_t st) {
  if (n <= 0) return 0;
  S2ST_LAUNCH(scale_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, a, y, n);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

namespace {
// The generator's stop rule on the device (speech_generator_for_s2st.py:88-99), one workgroup: after step `step`
//   cur_finished = eos_prob > thr ; out_lens[~finished & cur_finished] = step + 1 ; finished |= cur_finished
// and the key lengths of the NEXT step's self-attention, cur_out_lens of :84-85 (out_lens with the "still running" value
// max_iter replaced by step + 2).  n_done[step] = number of finished utterances after this step: the host reads it a few
// steps late (no synchronisation inside the loop) and discards the steps it ran past the stop.
__global__ __launch_bounds__(256) void decode_stop_update_kernel(const float* __restrict__ eos_prob, float thr, int step,
                                                                 int max_iter, int B, int* __restrict__ finished,
                                                                 int* __restrict__ out_lens, int* __restrict__ klen_next,
                                                                 int* __restrict__ n_done) {
  __shared__ int cnt[4];
  int c = 0;
  for (int b = threadIdx.x; b < B; b += 256) {
    const int cur = eos_prob[b] > thr ? 1 : 0;
    int f = finished[b], ol = out_lens[b];
    if (!f && cur) ol = step + 1;
    f |= cur;
    finished[b] = f;
    out_lens[b] = ol;
    klen_next[b] = ol == max_iter ? step + 2 : ol;
    c += f;
  }
  c = (int)wave_sum((float)c);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) n_done[step] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
}

// ---- replayable decode step (engine.cpp, include/s2st_hip.h s2st_decode_replay) ---------------------------------------
// Seeds as the host derives them: decode_step gets seed0 + step, its i-th dropout site next_seed() = seed * 0x100000001B3 +
// (i + 1) * 0x9E3779B97F4A7C15 (engine.cpp next_seed; site counts from 1).
__device__ __forceinline__ uint64_t replay_site_seed(uint64_t seed0, int step, int site) {
  return (seed0 + (uint64_t)step) * 0x100000001B3ULL + (uint64_t)(site + 1) * 0x9E3779B97F4A7C15ULL;
}
__global__ __launch_bounds__(1024) void decode_replay_init_kernel(int* __restrict__ step, uint64_t* __restrict__ seeds,
                                                                  float* __restrict__ cur_feat, long n_feat,
                                                                  float* __restrict__ pe_cur, const float* __restrict__ pe_alpha,
                                                                  int Cd, uint64_t seed0) {
  const int tid = threadIdx.x;
  for (long i = tid; i < n_feat; i += 1024) cur_feat[i] = 0.f;  // the zero frame (speech_generator_for_s2st.py:76-77)
  for (int c = tid; c < Cd; c += 1024) pe_cur[c] = pe_alpha[2L * Cd + c];
  if (tid < 8) seeds[tid] = replay_site_seed(seed0, 0, tid);
  if (tid == 0) *step = 0;
}
// Commit of a replayed step.  Workgroup 0: the stop rule of decode_stop_update_kernel on the step's probabilities (+ the
// step's row of eos_all), the next step's position row and seeds; the other workgroups: the step's features / alignment into
// row `step` of the run's buffers.  Every workgroup only READS the step counter: decode_replay_advance_kernel, the next node
// of the stream, increments it (one workgroup copying 37 k floats by itself cost 60 us per step).
__global__ __launch_bounds__(256) void decode_replay_commit_kernel(const int* __restrict__ step_p, uint64_t* __restrict__ seeds,
                                                                   const float* __restrict__ cur_feat,
                                                                   const float* __restrict__ cur_eos,
                                                                   const float* __restrict__ cur_attn, float* __restrict__ pe_cur,
                                                                   const float* __restrict__ pe_alpha, int pe_rows, int Cd,
                                                                   uint64_t seed0, float thr, int max_iter, int B, int out_dim, int E,
                                                                   int* __restrict__ finished, int* __restrict__ out_lens,
                                                                   int* __restrict__ klen_next, int* __restrict__ n_done,
                                                                   float* __restrict__ feat_all, float* __restrict__ eos_all,
                                                                   float* __restrict__ attn_all) {
  const int tid = threadIdx.x, step = *step_p;
  if (blockIdx.x == 0) {
    __shared__ int cnt[4];
    int c = 0;
    for (int b = tid; b < B; b += 256) {
      const float pr = cur_eos[b];
      const int cur = pr > thr ? 1 : 0;
      int f = finished[b], ol = out_lens[b];
      if (!f && cur) ol = step + 1;
      f |= cur;
      finished[b] = f;
      out_lens[b] = ol;
      klen_next[b] = ol == max_iter ? step + 2 : ol;
      c += f;
      if (step < max_iter) eos_all[(long)step * B + b] = pr;
    }
    c = (int)wave_sum((float)c);
    if ((tid & 63) == 0) cnt[tid >> 6] = c;
    const int row = min(step + 3, pe_rows - 1);  // next step's position: (step + 1) + 2
    for (int k = tid; k < Cd; k += 256) pe_cur[k] = pe_alpha[(long)row * Cd + k];
    if (tid < 8) seeds[tid] = replay_site_seed(seed0, step + 1, tid);
    __syncthreads();
    if (tid == 0 && step < max_iter) n_done[step] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    return;
  }
  if (step >= max_iter) return;
  const long nf = (long)B * out_dim, na = (attn_all && cur_attn) ? (long)B * E : 0;
  const long i = ((long)blockIdx.x - 1) * 256 + tid;
  if (i < nf) feat_all[(long)step * nf + i] = cur_feat[i];
  else if (i - nf < na) attn_all[(long)step * na + (i - nf)] = cur_attn[i - nf];
}
__global__ void decode_replay_advance_kernel(int* __restrict__ step_p) { *step_p += 1; }
}  // namespace

int s2st_decode_replay_init(int* step, uint64_t* seeds, float* cur_feat, long n_feat, float* pe_cur, const float* pe_alpha, int Cd,
                            uint64_t seed0, hipStream_t st) {
  if (!step || !seeds || !cur_feat || !pe_cur || !pe_alpha || n_feat <= 0 || Cd <= 0) return S2ST_ERR_ARG;
  S2ST_LAUNCH(decode_replay_init_kernel, dim3(1), dim3(1024), 0, st, step, seeds, cur_feat, n_feat, pe_cur, pe_alpha, Cd, seed0);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
int s2st_decode_replay_commit(int* step, uint64_t* seeds, const float* cur_feat, const float* cur_eos, const float* cur_attn,
                              float* pe_cur, const float* pe_alpha, int pe_rows, int Cd, uint64_t seed0, float thr, int max_iter,
                              int B, int out_dim, int E, int* finished, int* out_lens, int* klen_next, int* n_done,
                              float* feat_all, float* eos_all, float* attn_all, hipStream_t st) {
  if (B <= 0) return 0;
  if (!step || !seeds || !cur_feat || !cur_eos || !pe_cur || !pe_alpha || !finished || !out_lens || !klen_next || !n_done ||
      !feat_all || !eos_all)
    return S2ST_ERR_ARG;
  const long n_copy = (long)B * out_dim + ((attn_all && cur_attn) ? (long)B * E : 0);
  S2ST_LAUNCH(decode_replay_commit_kernel, dim3(1 + (unsigned)((n_copy + 255) / 256)), dim3(256), 0, st, (const int*)step, seeds,
              cur_feat, cur_eos, cur_attn, pe_cur, pe_alpha, pe_rows, Cd, seed0, thr, max_iter, B, out_dim, E, finished, out_lens,
              klen_next, n_done, feat_all, eos_all, attn_all);
  S2ST_LAUNCH(decode_replay_advance_kernel, dim3(1), dim3(1), 0, st, step);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_decode_stop_update(const float* eos_prob, float thr, int step, int max_iter, int B, int* finished, int* out_lens,
                            int* klen_next, int* n_done, hipStream_t st) {
  if (B <= 0) return 0;
  if (!eos_prob || !finished || !out_lens || !klen_next || !n_done || step < 0) return S2ST_ERR_ARG;
  S2ST_LAUNCH(decode_stop_update_kernel, dim3(1), dim3(256), 0, st, eos_prob, thr, step, max_iter, B, finished, out_lens, klen_next,
              n_done);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_sigmoid(const float* x, float* y, long n, hipStream_t st) {
  if (n <= 0) return 0;
  S2ST_LAUNCH(sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_argmax_dim1(const float* x, long* idx, int B, int E, int D, hipStream_t st) {
  const long n = (long)B * D;
  if (n <= 0 || E <= 0) return 0;
  S2ST_LAUNCH(argmax_dim1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, idx, B, E, D);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_affine_cols(const float* x, const float* scale, const float* shift, float* y, long rows, int C,
                     hipStream_t st) {
  const long n = rows * C;
  if (n <= 0) return 0;
  S2ST_LAUNCH(affine_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, scale, shift, y, rows,
                     C);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_exp_transpose(const float* x, float* y, int T, int C, hipStream_t st) {
  const long n = (long)T * C;
  if (n <= 0) return 0;
  S2ST_LAUNCH(exp_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, T, C);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
int s2st_clamp_min(float* x, long n, float lo, hipStream_t st) {
  if (n <= 0) return 0;
  S2ST_LAUNCH(clamp_min_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, lo);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_gl_polar(const float* mag, const float* ang, float* X, int F, int T, hipStream_t st) {
  const long n = (long)F * T;
  if (n <= 0) return 0;
  S2ST_LAUNCH(gl_polar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mag, ang, X, F, T);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_gl_project(const float* mag, const float* Y, float* X, int F, int T, hipStream_t st) {
  const long n = (long)F * T;
  if (n <= 0) return 0;
  S2ST_LAUNCH(gl_project_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mag, Y, X, F, T);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_reflect_pad(const float* x, float* y, int n, int pad, hipStream_t st) {
  if (n <= pad) return S2ST_ERR_SHAPE;
  const long m = (long)n + 2 * pad;
  S2ST_LAUNCH(reflect_pad_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, x, y, n, pad);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_gl_overlap_add(const float* frames, const float* wsq, float* wave, int T, int n_fft, int hop, int n_out,
                        hipStream_t st) {
  if (n_out <= 0) return 0;
  S2ST_LAUNCH(gl_overlap_add_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, frames, wsq, wave,
                     T, n_fft, hop, n_out, 1.1754944e-38f);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_gl_polar_split(const float* mag, const float* aux, int from_spectrum, const int* tl, uint16_t* Xs, int U, int F,
                        int Fp, int Tmax, hipStream_t st) {
  if (Fp % 4 || Fp < F) return S2ST_ERR_SHAPE;
  const long n = (long)U * Tmax * (Fp / 4);
  if (n <= 0) return 0;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (from_spectrum)
    S2ST_LAUNCH(gl_polar_split_kernel<1>, grid, dim3(256), 0, st, mag, aux, tl, Xs, U, F, Fp, Tmax);
  else
    S2ST_LAUNCH(gl_polar_split_kernel<0>, grid, dim3(256), 0, st, mag, aux, tl, Xs, U, F, Fp, Tmax);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
int s2st_gl_frame_split(const float* wave, const int* tl, uint16_t* As, int U, int Tmax, int hop, int n_fft, int Lw,
                        hipStream_t st) {
  if (n_fft % 4) return S2ST_ERR_SHAPE;
  const long n = (long)U * Tmax * (n_fft / 4);
  if (n <= 0) return 0;
  S2ST_LAUNCH(gl_frame_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wave, tl, As, U, Tmax,
                     hop, n_fft, Lw);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
int s2st_gl_overlap_add_b(const float* frames, const float* wsq_all, const long* wsq_off, const int* tl, float* wave,
                          int U, int Tmax, int n_fft, int hop, int Lw, hipStream_t st) {
  const long n = (long)U * Lw;
  if (n <= 0) return 0;
  S2ST_LAUNCH(gl_overlap_add_b_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, frames, wsq_all,
                     wsq_off, tl, wave, U, Tmax, n_fft, hop, Lw, 1.1754944e-38f);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

// ---- FFT-based Griffin-Lim launchers (n_fft a power of two in 256 ... 2048) ------------------------------------------
bool s2st_gl_fft_supported(int n_fft) { return n_fft == 256 || n_fft == 512 || n_fft == 1024 || n_fft == 2048; }

int s2st_gl_polar_c(const float* mag, const float* ang, const int* tl, float* X, int U, int F, int Tmax, hipStream_t st) {
  const long n = (long)U * Tmax * F;
  if (n <= 0) return 0;
  S2ST_LAUNCH(gl_polar_c_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mag, ang, tl, reinterpret_cast<cplx*>(X), U, F, Tmax);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

// x <- exp(x) in place (the vocoder's log-mel -> mel step, vocoder.py:139, for a whole padded batch at once)
int s2st_exp_inplace(float* x, long n, hipStream_t st) {
  if (n <= 0) return 0;
  S2ST_LAUNCH(exp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_gl_polar_u(const float* mag, const double* uni, const long* uoff, const int* tl, uint64_t seed, float* X, int U, int F,
                    int Tmax, hipStream_t st) {
  const long n = (long)U * F * Tmax;
  if (n <= 0) return 0;
  if (uni && !uoff) return S2ST_ERR_ARG;
  S2ST_LAUNCH(gl_polar_u_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mag, uni, uoff, tl, seed,
              reinterpret_cast<cplx*>(X), U, F, Tmax);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

namespace {
template <int N>
int gl_fft_launch(int inverse, const float* wave_or_frames, const int* tl, const float* win, const float* tw, const float* mag,
                  float* X, float* frames, int U, int Tmax, int hop, int Lw, hipStream_t st) {
  const long npairs = ((long)U * Tmax + 1) / 2;
  if (npairs <= 0) return 0;
  // (4 k workgroups at most: a workgroup keeps its twiddle table over the pairs it walks)
  const unsigned grid = (unsigned)(npairs < 4096 ? npairs : 4096);
  if (!inverse)
    S2ST_LAUNCH(gl_stft_project_kernel<N>, dim3(grid), dim3(256), 0, st, wave_or_frames, tl, win, reinterpret_cast<const cplx*>(tw),
                mag, reinterpret_cast<cplx*>(X), U, Tmax, hop, Lw, npairs);
  else
    S2ST_LAUNCH(gl_istft_frames_kernel<N>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const cplx*>(X), tl, win,
                reinterpret_cast<const cplx*>(tw), frames, U, Tmax, hop, npairs);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
}  // namespace

// wave [U][Lw] -> X [U * Tmax][n_fft / 2 + 1] complex (re, im interleaved), projected onto the magnitudes mag [U * Tmax][F]
int s2st_gl_stft_project(const float* wave, const int* tl, const float* win, const float* tw, const float* mag, float* X, int U,
                         int Tmax, int n_fft, int hop, int Lw, hipStream_t st) {
  switch (n_fft) {
    case 256: return gl_fft_launch<256>(0, wave, tl, win, tw, mag, X, nullptr, U, Tmax, hop, Lw, st);
    case 512: return gl_fft_launch<512>(0, wave, tl, win, tw, mag, X, nullptr, U, Tmax, hop, Lw, st);
    case 1024: return gl_fft_launch<1024>(0, wave, tl, win, tw, mag, X, nullptr, U, Tmax, hop, Lw, st);
    case 2048: return gl_fft_launch<2048>(0, wave, tl, win, tw, mag, X, nullptr, U, Tmax, hop, Lw, st);
  }
  return S2ST_ERR_SHAPE;
}

template <int N>
int gl_istft_ola_launch(const float* X, const int* tl, const float* win, const float* tw, const float* wsq_all, const long* wsq_off,
                        float* wave, int U, int Tmax, int hop, int Lw, hipStream_t st) {
  if (U <= 0 || Lw <= 0) return 0;
  if (hop < 1 || hop > N) return S2ST_ERR_ARG;
  // R frames per workgroup: about three workgroups per CU over the batch (S2ST_GL_OLA_FRAMES overrides), at least 16 so that
  // the ceil(N / hop) frames recomputed in front of each block stay a small share
  static const int r_env = [] {
    const char* e = s2st_env_str("S2ST_GL_OLA_FRAMES");
    return e ? atoi(e) : 0;
  }();
  int R = r_env > 0 ? r_env : (int)(((long)U * Tmax + 3 * 256 - 1) / (3 * 256));  // (256 CUs)
  if (R < 16 && r_env <= 0) R = 16;
  if (R > Tmax) R = Tmax;
  if (R < 1) R = 1;
  const int S = R * hop;
  const int C = N + 2 * hop <= 2 * N ? 2 * N : 4 * N;
  const int lds = 2 * FftLds<N>::SIZE * (int)sizeof(cplx) + C * (int)sizeof(float);
  static int configured = 0;  // (per instantiation)
  if (lds > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gl_istft_ola_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
        hipSuccess)
      return S2ST_ERR_LAUNCH;
    configured = lds;
  }
  const float tiny = 1.1754944e-38f;
  S2ST_LAUNCH(gl_istft_ola_kernel<N>, dim3((unsigned)((Lw + S - 1) / S), (unsigned)U), dim3(256), lds, st, reinterpret_cast<const cplx*>(X),
              tl, win, reinterpret_cast<const cplx*>(tw), wsq_all, wsq_off, wave, Tmax, hop, Lw, S, C, tiny);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
// X [U * Tmax][F] complex -> wave [U][Lw]: inverse transforms, overlap-add, window-sum-square normalisation, trim (one launch)
int s2st_gl_istft_ola(const float* X, const int* tl, const float* win, const float* tw, const float* wsq_all, const long* wsq_off,
                      float* wave, int U, int Tmax, int n_fft, int hop, int Lw, hipStream_t st) {
  switch (n_fft) {
    case 256: return gl_istft_ola_launch<256>(X, tl, win, tw, wsq_all, wsq_off, wave, U, Tmax, hop, Lw, st);
    case 512: return gl_istft_ola_launch<512>(X, tl, win, tw, wsq_all, wsq_off, wave, U, Tmax, hop, Lw, st);
    case 1024: return gl_istft_ola_launch<1024>(X, tl, win, tw, wsq_all, wsq_off, wave, U, Tmax, hop, Lw, st);
    case 2048: return gl_istft_ola_launch<2048>(X, tl, win, tw, wsq_all, wsq_off, wave, U, Tmax, hop, Lw, st);
  }
  return S2ST_ERR_ARG;
}
// X [U * Tmax][F] complex -> frames [U * Tmax][n_fft] (windowed synthesis frames; s2st_gl_overlap_add_b finishes)
int s2st_gl_istft_frames(const float* X, const int* tl, const float* win, const float* tw, float* frames, int U, int Tmax,
                         int n_fft, int hop, hipStream_t st) {
  switch (n_fft) {
    case 256: return gl_fft_launch<256>(1, nullptr, tl, win, tw, nullptr, const_cast<float*>(X), frames, U, Tmax, hop, 0, st);
    case 512: return gl_fft_launch<512>(1, nullptr, tl, win, tw, nullptr, const_cast<float*>(X), frames, U, Tmax, hop, 0, st);
    case 1024: return gl_fft_launch<1024>(1, nullptr, tl, win, tw, nullptr, const_cast<float*>(X), frames, U, Tmax, hop, 0, st);
    case 2048: return gl_fft_launch<2048>(1, nullptr, tl, win, tw, nullptr, const_cast<float*>(X), frames, U, Tmax, hop, 0, st);
  }
  return S2ST_ERR_SHAPE;
}
