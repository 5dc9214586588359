// s2st_loss kernels: masked L1 + MSE (pre/post-net) + BCE-with-logits stop loss,
// label-smoothed cross entropy fused with log-softmax and accuracy, CTC (log-softmax,
// alpha/beta recursions in LDS, gradient w.r.t. the logits).
//
// Reference call sites replaced: examples/s2s_trans/criterions/s2st_loss.py:294-315
// (compute_loss), :33-50 + :330-348 (label_smoothed_nll_loss, accuracy), :229-243 with
// s2st_transformer.py:458-463 (log_softmax + torch.nn.CTCLoss(mean, zero_infinity)).
#include "s2st_ops.h"
#include "s2st_prof.h"

namespace {

__device__ __forceinline__ float softplusf_(float x) {  // log(1 + exp(x)), stable
  return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}

// The workgroups' sums of a loss kernel, K <= 4 per workgroup.  ordered != null: every workgroup stores its sums at
// ordered[k][blockIdx.x]; loss_finalize_kernel adds them in workgroup order (the logged losses repeat bit for bit; a
// first form -- the last workgroup to arrive adds them, found through a returning atomic counter -- cost 20 - 40 us per
// launch: ~1 k returning atomics on one address).  ordered == null (the stand-alone C-ABI calls): float atomics.
__device__ __forceinline__ void block_stats_out(float* __restrict__ stats, int K, const float (*red)[4],
                                                float* __restrict__ ordered) {
  const int tid = threadIdx.x;
  if (tid >= K) return;
  const float v = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
  if (ordered) ordered[(long)tid * gridDim.x + blockIdx.x] = v;
  else atomicAdd(&stats[tid], v);
}

// One thread per (row, f) element for the mel terms; thread f == 0 of each row also does the
// stop-token term.  stats: [0] sum |fo-t| + |fp-t|, [1] sum (fo-t)^2 + (fp-t)^2, [2] sum bce
__global__ __launch_bounds__(256) void mel_loss_kernel(
    const float* __restrict__ feat, const float* __restrict__ post, const float* __restrict__ eos,
    const float* __restrict__ tgt, const int* __restrict__ lens, int B, int D, int F,
    float pos_weight, float* __restrict__ stats, float c_l1, float c_mse, float c_eos,
    float* __restrict__ dfeat, float* __restrict__ dpost, float* __restrict__ deos, float* __restrict__ ordered) {
  __shared__ float red[3][4];
  const long n = (long)B * D * F;
  float a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    long row = i / F;
    int f = (int)(i - row * F);
    int b = (int)(row / D), t = (int)(row - (long)b * D);
    int len = lens[b];
    bool valid = t < len;
    float gt = tgt[i];
    float e1 = feat[i] - gt, e2 = post[i] - gt;
    if (valid) {
      a1 += fabsf(e1) + fabsf(e2);
      a2 += e1 * e1 + e2 * e2;
    }
    if (dfeat) {
      float s1 = e1 > 0.f ? 1.f : (e1 < 0.f ? -1.f : 0.f);
      float s2 = e2 > 0.f ? 1.f : (e2 < 0.f ? -1.f : 0.f);
      dfeat[i] = valid ? c_l1 * s1 + c_mse * 2.f * e1 : 0.f;
      dpost[i] = valid ? c_l1 * s2 + c_mse * 2.f * e2 : 0.f;
    }
    if (f == 0) {
      float x = eos[row];
      float y = (t == len - 1) ? 1.f : 0.f;
      if (valid) a3 += pos_weight * y * softplusf_(-x) + (1.f - y) * softplusf_(x);
      if (deos) {
        float sg = 1.f / (1.f + expf(-x));
        deos[row] = valid ? c_eos * (-pos_weight * y * (1.f - sg) + (1.f - y) * sg) : 0.f;
      }
    }
  }
  if (stats) {
    a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = a1; red[1][w] = a2; red[2][w] = a3; }
    __syncthreads();
    block_stats_out(stats, 3, red, ordered);
  }
}

// one wave per row
__global__ __launch_bounds__(256) void ls_ce_kernel(const float* __restrict__ logits,
                                                    const long* __restrict__ target, int rows,
                                                    int V, long pad, float eps,
                                                    float* __restrict__ stats,
                                                    float* __restrict__ dlogits, float gscale,
                                                    float* __restrict__ ordered) {
  __shared__ float red[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float nll_a = 0.f, sm_a = 0.f, cor_a = 0.f, tot_a = 0.f;
  const float eps_i = eps / (float)(V - 1);
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float* x = logits + (long)row * V;
    const long tg = target[row];
    float mx = -INFINITY;
    int am = 0;
    for (int c = lane; c < V; c += 64) {
      float v = x[c];
      if (v > mx) { mx = v; am = c; }
    }
    // wave arg-max: larger value wins, ties -> smaller index (torch.argmax: first maximum)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      float ov = __shfl_xor(mx, m);
      int oi = __shfl_xor(am, m);
      if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
    }
    float se = 0.f, sx = 0.f;
    for (int c = lane; c < V; c += 64) {
      se += expf(x[c] - mx);
      sx += x[c];
    }
    se = wave_sum(se);
    sx = wave_sum(sx);
    const float lse = logf(se) + mx;
    const bool valid = tg != pad;
    if (valid) {
      nll_a += lse - x[tg];          // every lane holds the same value; lane 0 is used below
      sm_a += (float)V * lse - sx;   // -sum_c lprob_c
      cor_a += (am == (int)tg) ? 1.f : 0.f;
      tot_a += 1.f;
    }
    if (dlogits) {
      float* d = dlogits + (long)row * V;
      for (int c = lane; c < V; c += 64) {
        float p = expf(x[c] - lse);
        float g = (1.f - eps - eps_i) * (p - (c == (int)tg ? 1.f : 0.f)) + eps_i * ((float)V * p - 1.f);
        d[c] = valid ? gscale * g : 0.f;
      }
    }
  }
  if (stats) {
    if (lane == 0) { red[0][wave] = nll_a; red[1][wave] = sm_a; red[2][wave] = cor_a; red[3][wave] = tot_a; }
    __syncthreads();
    block_stats_out(stats, 4, red, ordered);
  }
}

constexpr float CTC_NEG = -1.0e30f;
constexpr int CTC_MAXS = 2048;  // 2 * Lmax + 1 <= CTC_MAXS
constexpr int CTC_MAXV = 512;

__device__ __forceinline__ float lse3(float a, float b, float c) {
  float m = fmaxf(a, fmaxf(b, c));
  if (m < -1.0e29f) return CTC_NEG;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// CTC in three launches (round 2: one kernel, alpha sweep then beta sweep with LDS-atomic occupancies, 0.3 - 0.6 ms;
// round 3 first form: one kernel with both sweeps in one loop, still 0.46 ms -- every sequential step waited for a
// dependent global load behind a full __syncthreads, and the log-softmax and gradient phases ran on B of the 256 CUs):
//   1. log-softmax of every frame: log_softmax_rows_kernel, one wave per frame over the whole chip;
//   2. ctc_ab_kernel, one workgroup per utterance: alpha (waves 0, 1: forward in time) and beta (waves 2, 3: backward)
//      recursions in the SAME loop.  A thread owns the same <= PER states at every step, so their labels and "may skip"
//      flags sit in registers and the NEXT step's emissions lp_t(ext s) are fetched one step ahead; a step is LDS reads,
//      three exp + one log, one LDS write, s_waitcnt lgkmcnt(0) and a raw s_barrier -- no wait for global memory (the
//      alpha / beta rows go out as stores nobody waits for);
//   3. ctc_grad_kernel, one workgroup per CTC_FR frames of an utterance (the whole chip again), one wave per frame:
//      e[s] = exp(alpha_t(s) + beta_t(s) - lp_t(ext s) - ll) for all states into LDS, then lane v adds the e[s] of ITS
//      label's states in state order (a CSR list built per workgroup) and the blank's as a fixed-shape wave reduction
//      -- no atomics: the occupancies, hence the logits' gradient, repeat bit for bit.
// Workspace per utterance: alpha | beta rows [2][E][ss]; behind all of them one log-likelihood per utterance.
template <int PER>
__global__ __launch_bounds__(256) void ctc_ab_kernel(const float* __restrict__ lp_all, const long* __restrict__ targets,
                                                     int Lmax, const int* __restrict__ in_lens,
                                                     const int* __restrict__ tgt_lens, int E, int V,
                                                     float* __restrict__ ws_all, int ss, float* __restrict__ ll_all,
                                                     float* __restrict__ loss_out) {
  HIP_DYNAMIC_SHARED(unsigned char, smem_raw)
  float* rows = reinterpret_cast<float*>(smem_raw);  // alpha row pair [2][ss] | beta row pair [2][ss]
  int* ext = reinterpret_cast<int*>(rows + 4 * ss);  // [ss] extended label sequence
  const int b = blockIdx.x, tid = threadIdx.x;
  const int L = tgt_lens[b], Tb = min(in_lens[b], E), S = 2 * L + 1;
  const float* lp = lp_all + (long)b * E * V;
  const bool isA = tid < 128;
  const int lt = tid & 127, d = isA ? -1 : 1;
  float* mine = rows + (isA ? 0 : 2 * ss);                                // this direction's row pair
  float* out = ws_all + (long)b * 2 * E * ss + (isA ? 0 : (long)E * ss);  // this direction's rows in HBM
  for (int s = tid; s < S; s += 256) ext[s] = (s & 1) ? (int)targets[(long)b * Lmax + (s >> 1)] : 0;
  __syncthreads();
  // A thread's states: s_i = lt + 128 i, clamped to the last state -- a thread past the end repeats state S - 1
  // exactly (same reads, same value, same addresses written), which keeps the step free of branches
  int sx[PER], es[PER];
  bool has1[PER], has2[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int s = min(lt + 128 * i, S - 1);
    sx[i] = s;
    es[i] = ext[s];
    const int s1 = s + d, s2 = s + 2 * d;  // the neighbours this state is reached from (alpha) / leads to (beta)
    has1[i] = s1 >= 0 && s1 < S;
    has2[i] = (s & 1) && s2 >= 0 && s2 < S && ext[s2] != ext[s];
  }
  float em0[PER], em1[PER];
  if (Tb > 0) {  // first row: alpha_0 / beta_{T-1}
    const int t = isA ? 0 : Tb - 1;
    float* cu = mine + (t & 1) * ss;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = sx[i];
      const bool open = isA ? s < 2 : s >= S - 2;
      const float a = open ? lp[(long)t * V + es[i]] : CTC_NEG;
      cu[s] = a;
      out[(long)t * ss + s] = a;
    }
  }
  if (Tb > 1) {
    const int t = isA ? 1 : Tb - 2;
#pragma unroll
    for (int i = 0; i < PER; ++i) em0[i] = lp[(long)t * V + es[i]];
  }
  __syncthreads();
  // one step: emc = this step's emissions (fetched one step ago), emx receives the next step's (the frame index is
  // clamped at the last step: the loads are unconditional, the step is straight-line code)
  auto step = [&](int k, const float (&emc)[PER], float (&emx)[PER]) {
    const int t = isA ? k : Tb - 1 - k;
    const long tn = isA ? min(k + 1, Tb - 1) : max(Tb - 2 - k, 0);
#pragma unroll
    for (int i = 0; i < PER; ++i) emx[i] = lp[tn * V + es[i]];
    const float* pr = mine + ((t & 1) ^ 1) * ss;
    float* cu = mine + (t & 1) * ss;
    float a[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = sx[i];
      const float a0 = pr[s];
      const float a1 = has1[i] ? pr[s + d] : CTC_NEG;
      const float a2 = has2[i] ? pr[s + 2 * d] : CTC_NEG;
      a[i] = lse3(a0, a1, a2);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const float v = a[i] < -1.0e29f ? CTC_NEG : a[i] + emc[i];
      cu[sx[i]] = v;
      out[(long)t * ss + sx[i]] = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): my LDS writes are done; global traffic stays in flight
    __builtin_amdgcn_s_barrier();
  };
  int k = 1;
  for (; k + 1 < Tb; k += 2) {
    step(k, em0, em1);
    step(k + 1, em1, em0);
  }
  if (k < Tb) step(k, em0, em1);
  if (tid == 0) {
    float ll = CTC_NEG;
    if (Tb > 0) {
      const float* last = rows + ((Tb - 1) & 1) * ss;
      ll = lse3(last[S - 1], S >= 2 ? last[S - 2] : CTC_NEG, CTC_NEG);
    }
    ll_all[b] = ll;
    loss_out[b] = (ll < -1.0e29f) ? 0.f : -ll / (float)max(L, 1);  // zero_infinity
  }
}

constexpr int CTC_FR = 16;  // frames per workgroup of the gradient kernel
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ lp_all, const long* __restrict__ targets,
                                                       int Lmax, const int* __restrict__ in_lens,
                                                       const int* __restrict__ tgt_lens, int E, int V,
                                                       const float* __restrict__ ws_all, int ss,
                                                       const float* __restrict__ ll_all, float* __restrict__ dlogits,
                                                       float gscale) {
  HIP_DYNAMIC_SHARED(unsigned char, smem_raw)
  float* erow = reinterpret_cast<float*>(smem_raw);                        // [4][ss] per-wave occupancy terms
  int* ext = reinterpret_cast<int*>(erow + 4 * ss);                        // [ss]
  int* start = ext + ss;                                                   // [V + 1] states of label v: idx[start[v] .. start[v+1])
  unsigned short* idx = reinterpret_cast<unsigned short*>(start + V + 1);  // [ss]
  const int b = blockIdx.y, f0 = blockIdx.x * CTC_FR, f1 = min(f0 + CTC_FR, E);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = tgt_lens[b], Tb = min(in_lens[b], E), S = 2 * L + 1;
  const float ll = ll_all[b];
  float* dl = dlogits + (long)b * E * V;
  if (!(ll > -1.0e29f) || f0 >= Tb) {  // infeasible utterance (zero_infinity) or frames past its input length
    for (long i = (long)f0 * V + tid; i < (long)f1 * V; i += 256) dl[i] = 0.f;
    return;
  }
  const float* lp = lp_all + (long)b * E * V;
  const float* alpha = ws_all + (long)b * 2 * E * ss;
  const float* beta = alpha + (long)E * ss;
  const float sc = gscale / (float)max(L, 1);
  for (int s = tid; s < S; s += 256) ext[s] = (s & 1) ? (int)targets[(long)b * Lmax + (s >> 1)] : 0;
  for (int v = tid; v <= V; v += 256) start[v] = 0;
  __syncthreads();
  // the label states (odd s) per label, in state order: counts (integer LDS atomics), offsets (one wave scans), lists
  for (int s = 2 * tid + 1; s < S; s += 512) atomicAdd(&start[ext[s] + 1], 1);
  __syncthreads();
  if (wave == 0) {
    const int ch = (V + 63) / 64, v0 = lane * ch;
    int own = 0;
    for (int q = 0; q < ch; ++q) own += (v0 + q < V) ? start[v0 + q + 1] : 0;
    int incl = own;
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl(incl, max(lane - o, 0));
      if (lane >= o) incl += up;
    }
    int run = incl - own;
    for (int q = 0; q < ch; ++q)
      if (v0 + q < V) { run += start[v0 + q + 1]; start[v0 + q + 1] = run; }
  }
  __syncthreads();
  for (int s = 2 * tid + 1; s < S; s += 512) {
    const int v = ext[s];
    int rank = 0;
    for (int q = 1; q < s; q += 2) rank += ext[q] == v ? 1 : 0;
    idx[start[v] + rank] = (unsigned short)s;
  }
  __syncthreads();
  float* er = erow + wave * ss;
  for (int t0 = f0; t0 < f1; t0 += 4) {  // uniform over the workgroup: barriers inside
    const int t = t0 + wave;
    const bool on = t < Tb;
    const float* lpt = lp + (long)(on ? t : 0) * V;
    if (on) {
      for (int s = lane; s < S; s += 64) {
        const float al = alpha[(long)t * ss + s], be = beta[(long)t * ss + s];
        er[s] = (al > -1.0e29f && be > -1.0e29f) ? expf(al + be - lpt[ext[s]] - ll) : 0.f;
      }
    }
    __syncthreads();
    if (on) {
      float blank = 0.f;
      for (int s = 2 * lane; s < S; s += 128) blank += er[s];
      blank = wave_sum(blank);
      for (int v = lane; v < V; v += 64) {
        float o = v == 0 ? blank : 0.f;
        for (int q = start[v]; q < start[v + 1]; ++q) o += er[idx[q]];
        dl[(long)t * V + v] = sc * (expf(lpt[v]) - o);
      }
    } else if (t < f1) {
      for (int v = lane; v < V; v += 64) dl[(long)t * V + v] = 0.f;
    }
    __syncthreads();
  }
}

// stats[16..22] = {loss, l1, mse, eos, ctc, asr, st} from the raw sums (s2st_loss.py:245-257)
__global__ void loss_finalize_kernel(float* __restrict__ stats, const float* __restrict__ ctc_per, int B,
                                     float nf, float nr, float w_l1, float w_mse, float w_eos,
                                     float w_ctc, float w_asr, float w_st, float eps, int Vs, int Vt,
                                     float src_ntok, float tgt_ntok, const float* __restrict__ ctc_tgt_per,
                                     float w_ctc_tgt, s2st_loss_parts parts) {
  if (parts.on) {  // the loss kernels' per-workgroup sums, added in workgroup order (wave-strided, fixed tree)
    const int lane = threadIdx.x & 63;
    for (int o = threadIdx.x >> 6; o < 11; o += blockDim.x >> 6) {
      const int gi = o < 3 ? 0 : (o < 7 ? 1 : 2), k = o < 3 ? o : (o < 7 ? o - 3 : o - 7);
      const float* p = parts.part[gi];
      const int nb = parts.nblocks[gi];
      if (!p || nb <= 0) continue;
      float v = 0.f;
      for (int b = lane; b < nb; b += 64) v += p[(long)k * nb + b];
      v = wave_sum(v);
      if (lane == 0) stats[o] = v;
    }
    __syncthreads();
  }
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float l1 = w_l1 * stats[S2ST_STAT_L1_SUM] / nf;
  float mse = w_mse * stats[S2ST_STAT_MSE_SUM] / nf;
  float eos = w_eos * stats[S2ST_STAT_BCE_SUM] / nr;
  float ctc = 0.f;
  if (ctc_per) {
    for (int b = 0; b < B; ++b) ctc += ctc_per[b];
    ctc = w_ctc * ctc / (float)B;
  }
  float ctc_tgt = 0.f;  // second CTC head of the mtl variant (s2st_loss_mtl.py:171-186)
  if (ctc_tgt_per) {
    for (int b = 0; b < B; ++b) ctc_tgt += ctc_tgt_per[b];
    ctc_tgt = w_ctc_tgt * ctc_tgt / (float)B;
  }
  float asr = 0.f, st = 0.f;
  if (w_asr > 0.f && src_ntok > 0.f) {
    float ei = eps / (float)(Vs - 1);
    asr = w_asr * ((1.f - eps - ei) * stats[S2ST_STAT_ASR_NLL] + ei * stats[S2ST_STAT_ASR_SMOOTH]) / src_ntok;
  }
  if (w_st > 0.f && tgt_ntok > 0.f) {
    float ei = eps / (float)(Vt - 1);
    st = w_st * ((1.f - eps - ei) * stats[S2ST_STAT_ST_NLL] + ei * stats[S2ST_STAT_ST_SMOOTH]) / tgt_ntok;
  }
  stats[S2ST_STAT_L1] = l1;
  stats[S2ST_STAT_MSE] = mse;
  stats[S2ST_STAT_EOS] = eos;
  stats[S2ST_STAT_CTC] = ctc;
  stats[S2ST_STAT_ASR] = asr;
  stats[S2ST_STAT_ST] = st;
  stats[S2ST_STAT_CTC_TGT] = ctc_tgt;
  stats[S2ST_STAT_LOSS] = l1 + mse + eos + ctc + ctc_tgt + asr + st;
}

// (log-)softmax over the last dimension, one wave per row (models' get_normalized_probs: s2st_transformer.py:458-463,
// fairseq/utils.py log_softmax / softmax in fp32)
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y,
                                                               long ldy, int rows, int V, int log_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, xr[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += expf(xr[c] - mx);
  s = wave_sum(s);
  const float ls = logf(s);
  float* yr = y + (long)row * ldy;
  for (int c = lane; c < V; c += 64) {
    const float v = xr[c] - mx - ls;
    yr[c] = log_out ? v : expf(v);
  }
}

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH)

int s2st_mel_loss(const float* feat, const float* post, const float* eos, const float* tgt,
                  const int* lens, int B, int D, int F, float pos_weight, float* stats, float c_l1,
                  float c_mse, float c_eos, float* dfeat, float* dpost, float* deos,
                  hipStream_t st, float* ordered, int* nblocks_out) {
  long n = (long)B * D * F;
  if (n <= 0) return 0;
  long blocks = (n + 256 * 4 - 1) / (256 * 4);
  if (blocks > 2048) blocks = 2048;
  S2ST_LAUNCH(mel_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, st, feat, post, eos, tgt,
                     lens, B, D, F, pos_weight, stats, c_l1, c_mse, c_eos, dfeat, dpost, deos, ordered);
  if (nblocks_out) *nblocks_out = (int)blocks;
  return LAUNCH_OK();
}

int s2st_ls_ce(const float* logits, const long* target, int rows, int V, long pad, float eps,
               float* stats, float* dlogits, float gscale, hipStream_t st, float* ordered, int* nblocks_out) {
  if (rows <= 0) return 0;
  int blocks = (rows + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  S2ST_LAUNCH(ls_ce_kernel, dim3(blocks), dim3(256), 0, st, logits, target, rows, V, pad, eps,
                     stats, dlogits, gscale, ordered);
  if (nblocks_out) *nblocks_out = blocks;
  return LAUNCH_OK();
}

long s2st_ctc_workspace_floats(int B, int E, int Lmax) {
  long ss = ((2L * Lmax + 1 + 3) / 4) * 4;
  return 2L * B * E * ss + B;  // alpha and beta rows; one log-likelihood per utterance
}

int s2st_ctc(const float* logits, const long* targets, int Lmax, const int* in_lens,
             const int* tgt_lens, int B, int E, int V, float* lprobs, float* loss_per_utt,
             float* dlogits, float gscale, float* ws, hipStream_t st) {
  if (B <= 0) return 0;
  if (2 * Lmax + 1 > CTC_MAXS || V > CTC_MAXV || E <= 0) return S2ST_ERR_SHAPE;
  const int S = 2 * Lmax + 1, ss = ((S + 3) / 4) * 4;
  float* ll = ws + 2L * B * E * ss;
  S2ST_LAUNCH(log_softmax_rows_kernel, dim3((B * E + 3) / 4), dim3(256), 0, st, logits, (long)V, lprobs, (long)V, B * E, V, 1);
  const unsigned lds_ab = (unsigned)(ss * 5 * 4);
#define CTC_AB(PER) S2ST_LAUNCH(ctc_ab_kernel<PER>, dim3(B), dim3(256), lds_ab, st, lprobs, targets, Lmax, in_lens, tgt_lens, E, V, ws, ss, ll, loss_per_utt)
  if (S <= 256) CTC_AB(2);
  else if (S <= 512) CTC_AB(4);
  else CTC_AB(16);
#undef CTC_AB
  if (dlogits) {
    const unsigned lds_g = (unsigned)(ss * (4 * 4 + 4 + 2) + (V + 1) * 4 + 16);
    S2ST_LAUNCH(ctc_grad_kernel, dim3((E + CTC_FR - 1) / CTC_FR, B), dim3(256), lds_g, st, lprobs, targets, Lmax, in_lens,
                tgt_lens, E, V, ws, ss, ll, dlogits, gscale);
  }
  return LAUNCH_OK();
}

int s2st_loss_finalize(float* stats, const float* ctc_per, int B, float nf, float nr, float w_l1,
                       float w_mse, float w_eos, float w_ctc, float w_asr, float w_st, float eps, int Vs,
                       int Vt, float src_ntok, float tgt_ntok, hipStream_t st, const float* ctc_tgt_per, float w_ctc_tgt,
                       const s2st_loss_parts* parts) {
  s2st_loss_parts pt{};
  if (parts) { pt = *parts; pt.on = 1; }
  S2ST_LAUNCH(loss_finalize_kernel, dim3(1), dim3(parts ? 256 : 64), 0, st, stats, ctc_per, B, nf, nr, w_l1, w_mse,
                     w_eos, w_ctc, w_asr, w_st, eps, Vs, Vt, src_ntok, tgt_ntok, ctc_tgt_per, w_ctc_tgt, pt);
  return LAUNCH_OK();
}

int s2st_log_softmax_rows(const float* x, long ldx, float* y, long ldy, int rows, int V, int log_out, hipStream_t st) {
  if (rows <= 0 || V <= 0) return 0;
  S2ST_LAUNCH(log_softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, y, ldy, rows, V, log_out);
  return hipGetLastError() == hipSuccess ? 0 : S2ST_ERR_LAUNCH;
}
