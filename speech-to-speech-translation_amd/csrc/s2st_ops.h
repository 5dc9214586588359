// Host-side launcher interface of the s2st HIP kernels (internal C++ API; the C ABI in
// include/s2st_hip.h wraps these).  All launchers are stateless, stream-ordered, take
// caller-owned device buffers and return 0 or a negative error code (no exceptions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "s2st_common.h"

// ---------------------------------------------------------------------------------------
// GEMM:  C(m, n) = epilogue( alpha * sum_k A(m, k) * B(n, k) )      fp32 in HBM,
// bf16 MFMA (v_mfma_f32_16x16x32_bf16) with fp32 accumulation; `precise` selects the
// bf16x3 split (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, ~fp32 accuracy) used by parity tests.
// ---------------------------------------------------------------------------------------
typedef s2st_gemm_operand GemmOperand;
typedef s2st_gemm_out GemmOut;
typedef s2st_gemm_epilogue GemmEpilogue;
typedef s2st_gemm_args GemmArgs;

inline GemmOperand gemm_rowmajor(const float* p, long ld) {  // X[r][k], k contiguous
  GemmOperand o;
  o.p = p; o.kmajor = 1; o._pad = 0; o.sp.ld = ld; o.sp.bs = 0; o.sp.per = 0; o.sp._pad = 0;
  o.zo = o.zi = 0;
  return o;
}
inline GemmOperand gemm_colmajor(const float* p, long ld) {  // X stored [k][r], r contiguous
  GemmOperand o = gemm_rowmajor(p, ld);
  o.kmajor = 0;
  return o;
}
inline GemmOut gemm_out(float* p, long ld) {
  GemmOut o;
  o.p = p; o.sp.ld = ld; o.sp.bs = 0; o.sp.per = 0; o.sp._pad = 0; o.zo = o.zi = 0;
  return o;
}
inline GemmEpilogue gemm_epi_default() {
  GemmEpilogue e;
  e.alpha = 1.f; e.bias = nullptr; e.act = 0; e.drop_p = 0.f; e.seed = 0; e.resid = nullptr;
  e.accumulate = 0;
  return e;
}

int s2st_gemm(GemmArgs g, hipStream_t st);

// ---------------------------------------------------------------------------------------
// row ops (rowops.hip)
// ---------------------------------------------------------------------------------------
int s2st_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int rows, int cols, float eps, hipStream_t st);
// dx (+)= ...; dgamma/dbeta accumulated with atomics (caller zeroes or owns accumulation)
int s2st_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                       const float* rstd, float* dx, int dx_accumulate, float* dgamma,
                       float* dbeta, int rows, int cols, hipStream_t st);

// attention probabilities: p = softmax(s + masks) rows of [B, H, T, S(ld)]
//   key mask: col >= klen[b] -> -inf ; causal: col > row -> -inf
//   pd (optional, drop_p > 0) = dropout(p)
int s2st_softmax_fwd(const float* s, float* p, float* pd, const int* klen, int B, int H, int T,
                     int S, int ld, int causal, float drop_p, uint64_t seed, hipStream_t st);
// ds = p * (dp' - sum(dp' * p)),  dp' = dropmask * dpd ; in place allowed (ds == dpd)
int s2st_softmax_bwd(const float* p, const float* dpd, float* ds, int B, int H, int T, int S,
                     int ld, float drop_p, uint64_t seed, hipStream_t st);

// column sums: out[c] (+)= sum_r x[r][c]   (bias gradients)
int s2st_colsum(const float* x, long ld, int rows, int cols, float* out, int accumulate,
                hipStream_t st);
// mean over heads of attention probabilities: out[b][s][t] = mean_h p[b][h][t][s]
int s2st_attn_headmean(const float* p, float* out, int B, int H, int T, int S, int ld,
                       hipStream_t st);

// ---------------------------------------------------------------------------------------
// elementwise (elementwise.hip)
// ---------------------------------------------------------------------------------------
// y[r][c] = a[r][c] * sigmoid(a[r][c + C])   a: [rows][2C] ; y rows addressed via split
int s2st_glu_fwd(const float* a, float* y, Split ysp, int rows, int C, hipStream_t st);
int s2st_glu_bwd(const float* a, const float* dy, Split dysp, float* da, int rows, int C,
                 hipStream_t st);
// y[b][t][:] = scale * x[b][t][:] + alpha * PE(pos(b, t))  ; pos = t + 2 if t < len[b] else pad
// alpha read from device pointer if alpha_ptr != null (decoder pos_emb_alpha)
int s2st_add_pe(const float* x, float* y, const int* lens, int B, int T, int C, float scale,
                const float* alpha_ptr, float drop_p, uint64_t seed, hipStream_t st);
// text decoder variant: positions from token ids (pad = 1)
int s2st_embed_fwd(const long* tokens, const float* table, float* y, int B, int L, int Cin,
                   float scale, hipStream_t st);
int s2st_embed_bwd(const long* tokens, const float* dy, float* dtable, int B, int L, int Cin,
                   float scale, hipStream_t st);
int s2st_add_pe_tokens(const float* x, float* y, const long* tokens, int B, int L, int C,
                       float drop_p, uint64_t seed, hipStream_t st);
// dalpha += sum dy * PE ; used for decoder.pos_emb_alpha
int s2st_pe_alpha_bwd(const float* dy, const int* lens, int B, int T, int C, float* dalpha,
                      hipStream_t st);
// generic: y = x * dropmask (or y = x if p == 0), optional accumulate; flat n elements
int s2st_dropout(const float* x, float* y, long n, float p, uint64_t seed, int accumulate,
                 hipStream_t st);
// dz = dy * (y != 0 ? 1/(1-p) : 0)   (backward of dropout(relu(z)) given its OUTPUT y)
int s2st_relu_drop_bwd(const float* dy, const float* y, float* dz, long n, float p,
                       hipStream_t st);
int s2st_axpy(const float* x, float* y, long n, float a, hipStream_t st);  // y += a * x
int s2st_scale(float* x, long n, float a, hipStream_t st);
// copy rows [B][T][C] into a halo-padded buffer [B][T + 2*halo][C] (halo rows untouched)
int s2st_copy_rows(const float* x, Split xsp, float* y, Split ysp, int rows, int C,
                   hipStream_t st);
// conv weight W[O][I][Kw] -> Wf[O][Kw][I] (forward GEMM layout) and, if wd != null,
// Wd[I][Kw'][O] with Kw' = Kw-1-j (flipped; dgrad GEMM layout)
int s2st_conv_w_permute(const float* w, float* wf, float* wd, int O, int I, int Kw,
                        hipStream_t st);
// dW[O][I][Kw] += dWf[O][Kw][I]
int s2st_conv_w_unpermute_acc(const float* dwf, float* dw, int O, int I, int Kw, hipStream_t st);
// zero-stuff: up[b][2t][c] = x[b][t][c], odd rows zero (stride-2 conv dgrad)
// BatchNorm over rows (training): statistics, apply (+tanh)(+dropout), backward
int s2st_bn_stats(const float* x, int rows, int C, float* mean, float* var, float* run_mean,
                  float* run_var, float momentum, hipStream_t st);
int s2st_bn_apply(const float* x, const float* mean, const float* var, const float* gamma,
                  const float* beta, float* y, Split ysp, int rows, int C, float eps, int tanh_,
                  float drop_p, uint64_t seed, hipStream_t st);
int s2st_bn_bwd(const float* dy, Split dysp, const float* x, const float* y, Split ysp,
                const float* mean, const float* var, const float* gamma, float* dx,
                float* dgamma, float* dbeta, float* tmp2C, int rows, int C, float eps, int tanh_,
                float drop_p, uint64_t seed, hipStream_t st);

// ---------------------------------------------------------------------------------------
// losses (losses.hip)
// ---------------------------------------------------------------------------------------
// stats[0..4] += {sum|fo-t|, sum|fp-t|, sum(fo-t)^2, sum(fp-t)^2, sum bce}; valid rows only
// grads (if non-null) are d(total)/d(.) for total = wl1*(L1o+L1p)/Nf + wmse*(..)/Nf + weos*bce/Nr
int s2st_mel_loss(const float* feat, const float* post, const float* eos, const float* tgt,
                  const int* lens, int B, int D, int F, float pos_weight, float* stats,
                  hipStream_t st);
int s2st_mel_loss_bwd(const float* feat, const float* post, const float* eos, const float* tgt,
                      const int* lens, int B, int D, int F, float pos_weight, float w_l1,
                      float w_mse, float w_eos, float gscale, float* dfeat, float* dpost,
                      float* deos, hipStream_t st);
// label-smoothed CE over logits [rows][V]; stats += {nll_sum, smooth_sum, n_correct, total}
// dlogits = gscale * d(loss_sum)/dlogits with loss = (1-eps-eps_i)*nll + eps_i*smooth
int s2st_ls_ce(const float* logits, const long* target, int rows, int V, int pad, float eps,
               float* stats, float* dlogits, float gscale, hipStream_t st);
// log_softmax + CTC (blank 0).  logits [B][E][V].  ws: workspace >= B*E*(2L+1) floats * 2
int s2st_ctc(const float* logits, const long* targets /*[B][Lmax]*/, const int* in_lens,
             const int* tgt_lens, int B, int E, int V, int Lmax, float* lprobs_out,
             float* loss_per_utt, float* dlogits, float gscale_over_B, float* ws, hipStream_t st);

// ---------------------------------------------------------------------------------------
// optimizer (optim.hip)
// ---------------------------------------------------------------------------------------
int s2st_sumsq(const float* x, long n, float* out /* += */, hipStream_t st);
// g *= gmul * clip ; clip = min(1, max_norm / (sqrt(sumsq)*gmul + 1e-6)) ; fairseq Adam
int s2st_adam(float* p, float* g, float* m, float* v, long n, const float* sumsq, float gmul,
              float max_norm, float lr, float beta1, float beta2, float eps, float wd, int step,
              hipStream_t st);
