// Optimizer step over the flat parameter arena: gradient L2 norm, then one fused kernel for
// grad scaling (world / sample_size), clip-by-global-norm and the fairseq flavour of Adam.
//
// Reference call sites replaced: fairseq/trainer.py:838-873 (multiply_grads, clip_grad_norm,
// optimizer.step), fairseq/utils.py:345-395 (clip_grad_norm_; apex multi_tensor_l2norm),
// fairseq/optim/adam.py:163-239 (Adam.step; apex FusedAdam), fused_adam.py:11-37.
#include "s2st_ops.h"
#include "s2st_prof.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n,
                                                    float* __restrict__ out, int parts) {
  __shared__ float red[4];
  float a = 0.f;
  long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 v = x4[i];
    a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    a += x[i] * x[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s = red[0] + red[1] + red[2] + red[3];
    if (parts) out[blockIdx.x] = s;  // one partial per block, summed in index order by the consumer: no atomics
    else atomicAdd(out, s);
  }
}

// the gradient norm's sum of squares from the per-block partials of sumsq_kernel: every block of the consumer kernel
// forms the SAME sum in the same order (thread t adds parts[t], parts[t + 256], ...; wave butterfly; waves in order)
__device__ __forceinline__ float fold_sumsq_parts(const float* __restrict__ parts, int nparts) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) a += parts[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// gnorm = sqrt(sumsq) * gmul ; coef = max_norm > 0 ? min(1, max_norm / (gnorm + 1e-6)) : 1
// g' = g * gmul * coef ; m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2
// p -= lr*wd*p ; p -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
// ph (optional): bf16 copy of the updated parameters -- the GEMM operand copy the next forward would otherwise
// make with a separate pass over the arena
__device__ __forceinline__ float adam_one(float& g, float& m, float& v, float p, float coef, float b1, float b2,
                                          float eps, float wd_lr, float step_size) {
  g *= coef;
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  if (wd_lr != 0.f) p -= wd_lr * p;
  return p - step_size * m / (sqrtf(v) + eps);
}
// One element per thread and iteration, 1024 blocks: ~5.9 TB/s over the 32 (+2) bytes per parameter -- the kernel is
// HBM-bound; a float4 form, two / four elements in flight per thread and other grid sizes measured the same or slower
// (tools/adam_bench.py).  The four fp32 streams (every byte touched exactly once per update, 1.2 GB each way) are read and
// written with the NONTEMPORAL cache policy: the kernel alone gains 2 - 8 % and the step 0.03 - 0.06 ms on two boxes
// (profiles/r06_adam_nontemporal_ab.txt); the bf16 parameter copy, which the next forward reads, keeps the default policy.
// (The same policy on the GEMMs' operand loads, the weight transpose, the norm pass and the weight-gradient epilogues
// measured neutral to strongly negative: profiles/r06_dma_nontemporal_ab.txt, r06_nontemporal_other_streams_ab.txt.)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          long n, const float* __restrict__ sumsq,
                                                          float gmul, const float* __restrict__ gmul_dev,
                                                          float max_norm, float lr, float b1, float b2, float eps,
                                                          float wd, float step_size,
                                                          float* __restrict__ gnorm_out, uint16_t* __restrict__ ph,
                                                          int* __restrict__ skipped, int sumsq_parts, int zero_grad) {
  if (gmul_dev) gmul *= gmul_dev[0];
  const float gn = sqrtf(sumsq_parts > 0 ? fold_sumsq_parts(sumsq, sumsq_parts) : sumsq[0]) * gmul;
  if (gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) gnorm_out[0] = gn;
  if (!(gn < INFINITY)) {  // non-finite gradient norm: nothing is touched; the caller finds the count (trainer.py:860-867)
    if (skipped && blockIdx.x == 0 && threadIdx.x == 0) skipped[0] += 1;
    if (zero_grad)  // ... except the gradients, which the caller asked to find cleared for the next step
      for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) g[i] = 0.f;
    return;
  }
  float coef = gmul;
  if (max_norm > 0.f) coef *= fminf(1.f, max_norm / (gn + 1e-6f));
  const float wd_lr = wd * lr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gi = __builtin_nontemporal_load(g + i), mi = __builtin_nontemporal_load(m + i), vi = __builtin_nontemporal_load(v + i);
    const float pi = adam_one(gi, mi, vi, __builtin_nontemporal_load(p + i), coef, b1, b2, eps, wd_lr, step_size);
    __builtin_nontemporal_store(zero_grad ? 0.f : gi, g + i);
    __builtin_nontemporal_store(mi, m + i);
    __builtin_nontemporal_store(vi, v + i);
    __builtin_nontemporal_store(pi, p + i);
    if (ph) ph[i] = (uint16_t)(pack_bf16x4(pi, 0.f, 0.f, 0.f).x & 0xffffu);
  }
}

}  // namespace

static long sumsq_blocks(long n) {
  long blocks = (n / 4 + 255) / 256;
  if (blocks > S2ST_SUMSQ_PARTS) blocks = S2ST_SUMSQ_PARTS;
  return blocks < 1 ? 1 : blocks;
}
int s2st_sumsq(const float* x, long n, float* out, hipStream_t st) {
  if (n <= 0) return 0;
  s2st_launch("sumsq_kernel", 4.0 * n, 0.0, sumsq_kernel, dim3((unsigned)sumsq_blocks(n)), dim3(256), 0, st, x, n, out, 0);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
// parts[0 .. s2st_sumsq_nparts(n)) = per-block partial sums (plain stores: nothing to zero, nothing atomic); the
// count is what s2st_adam takes as sumsq_parts
int s2st_sumsq_parts(const float* x, long n, float* parts, hipStream_t st) {
  if (n <= 0) return 0;
  const long blocks = sumsq_blocks(n);
  s2st_launch("sumsq_kernel", 4.0 * n, 0.0, sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, parts, 1);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
long s2st_sumsq_nparts(long n) { return n <= 0 ? 0 : sumsq_blocks(n); }

int s2st_adam(float* p, float* g, float* m, float* v, long n, const float* sumsq, float gmul,
              const float* gmul_dev, float max_norm, float lr, float beta1, float beta2, float eps, float wd, int step,
              float* gnorm_out, hipStream_t st, uint16_t* ph, int* skipped, int sumsq_parts, int zero_grad) {
  if (n <= 0) return 0;
  double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  float step_size = (float)((double)lr * sqrt(bc2) / bc1);
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 || (ph && (uintptr_t)ph % 8)) return S2ST_ERR_ARG;
  long blocks = (n + 256 * 4 - 1) / (256 * 4);
  if (blocks > 1024) blocks = 1024;
  // bytes: p, g, m, v read and written (32 B per parameter) + the bf16 copy (2 B)
  s2st_launch("adam_kernel", (32.0 + (ph ? 2.0 : 0.0)) * n, 0.0, adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m,
              v, n, sumsq, gmul, gmul_dev, max_norm, lr, beta1, beta2, eps, wd, step_size, gnorm_out, ph, skipped, sumsq_parts, zero_grad);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

// ---- bf16 gradient exchange (runtime/distributed.py: --grad-exchange-dtype bf16) --------------------------------------
// A finished range of the gradient arena is rounded to bf16 (round-to-nearest-even, like every bf16 operand copy of the
// path), summed over the ranks in that type by the collective, and widened back into the fp32 arena: half the wire bytes
// of the default fp32 exchange (fairseq/models/distributed_fairseq_model.py:58-67 all-reduces fp32 gradients; opt-in).
namespace {
__global__ __launch_bounds__(256) void grad_pack_bf16_kernel(const float* __restrict__ g, uint16_t* __restrict__ out, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    reinterpret_cast<uint2*>(out)[i] = pack_bf16x4(v.x, v.y, v.z, v.w);
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const uint2 q = pack_bf16x4(g[i], 0.f, 0.f, 0.f);
    out[i] = (uint16_t)(q.x & 0xffffu);
  }
}
__global__ __launch_bounds__(256) void grad_unpack_bf16_kernel(const uint16_t* __restrict__ in, float* __restrict__ g, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const uint2 q = reinterpret_cast<const uint2*>(in)[i];
    float4 v;
    v.x = __uint_as_float(q.x << 16); v.y = __uint_as_float(q.x & 0xffff0000u);
    v.z = __uint_as_float(q.y << 16); v.w = __uint_as_float(q.y & 0xffff0000u);
    reinterpret_cast<float4*>(g)[i] = v;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    g[i] = __uint_as_float((uint32_t)in[i] << 16);
}
}  // namespace

// ---- 8-GPU pricing on ONE GPU: a stand-in for the collective's kernels --------------------------------
// A ring all-reduce over N ranks moves 2 (N - 1) / N of a bucket's bytes in and out of every GPU, through a handful of
// workgroups (RCCL's channels) that sit on CUs beside the backward and read / write HBM at the links' pace.  This kernel is
// that neighbour and nothing else: `wgs` workgroups read `move_bytes` from the bucket (cyclically) and write them to a
// scratch range, paced to `gbps` by the constant-frequency wall clock (each workgroup may move its share of the bytes no
// faster than gbps / wgs).  It changes no gradient.  bench.py --exchange-proxy launches it per bucket on the
// gradient-exchange stream behind both engine streams, where the real collective would go.
namespace {
__global__ __launch_bounds__(256) void exchange_proxy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n4,
                                                             long move4, double ticks_per_chunk) {
  // this workgroup's chunks: chunk c = 1024 float4 (16 KB); chunks are dealt round-robin to the workgroups
  const long nchunks = (move4 + 1023) / 1024;
  const unsigned long long t0 = wall_clock64();
  long done = 0;
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x, ++done) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long i = c * 1024 + q * 256 + threadIdx.x;
      if (i < move4) {
        const long j = i % n4;
        dst[j] = src[j];
      }
    }
    // pace: chunk number `done + 1` of this workgroup must not finish before its share of the wire time has passed
    const unsigned long long due = t0 + (unsigned long long)((double)(done + 1) * ticks_per_chunk);
    while (wall_clock64() < due) __builtin_amdgcn_s_sleep(8);
  }
}
}  // namespace

int s2st_exchange_proxy(const float* bucket, float* scratch, long n, long move_bytes, int wgs, float gbps, hipStream_t st) {
  if (n < 4 || move_bytes <= 0 || wgs <= 0) return 0;
  if (((uintptr_t)bucket % 16) || ((uintptr_t)scratch % 16) || gbps <= 0.f) return S2ST_ERR_ARG;
  const long n4 = n / 4, move4 = move_bytes / 16;
  // wall_clock64 ticks at 100 MHz on gfx9: a workgroup's 16 KB chunk is due every 16384 * wgs / (gbps * 1e9) s
  const double ticks_per_chunk = 16384.0 * wgs / ((double)gbps * 1e9) * 100e6;
  S2ST_LAUNCH(exchange_proxy_kernel, dim3((unsigned)wgs), dim3(256), 0, st, reinterpret_cast<const float4*>(bucket),
              reinterpret_cast<float4*>(scratch), n4, move4, ticks_per_chunk);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}

int s2st_grad_pack_bf16(const float* g, uint16_t* out, long n, hipStream_t st) {
  if (n <= 0) return 0;
  if (((uintptr_t)g % 16) || ((uintptr_t)out % 8)) return S2ST_ERR_ARG;
  long blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  S2ST_LAUNCH(grad_pack_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, out, n);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
int s2st_grad_unpack_bf16(const uint16_t* in, float* g, long n, hipStream_t st) {
  if (n <= 0) return 0;
  if (((uintptr_t)g % 16) || ((uintptr_t)in % 8)) return S2ST_ERR_ARG;
  long blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  S2ST_LAUNCH(grad_unpack_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, g, n);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
