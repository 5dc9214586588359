// extern "C" surface of libs2st_hip.so (see include/s2st_hip.h).
#include <hip/hip_ext.h>
#include "s2st_ops.h"

extern "C" {

int s2st_version(void) { return 100; }

int s2st_experimental_build(void) {
#ifdef S2ST_EXPERIMENTAL
  return 1;
#else
  return 0;
#endif
}

int s2st_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int s2st_exchange_proxy_f32(const float* bucket, float* scratch, int64_t n, int64_t move_bytes, int32_t wgs, float gbps, void* stream) {
  return s2st_exchange_proxy(bucket, scratch, (long)n, (long)move_bytes, wgs, gbps, (hipStream_t)stream);
}
int s2st_grad_pack_bf16_f32(const float* g, uint16_t* out, int64_t n, void* stream) {
  return s2st_grad_pack_bf16(g, out, (long)n, (hipStream_t)stream);
}

int s2st_grad_unpack_bf16_f32(const uint16_t* in, float* g, int64_t n, void* stream) {
  return s2st_grad_unpack_bf16(in, g, (long)n, (hipStream_t)stream);
}

int s2st_gemm_f32(const s2st_gemm_args* a, void* stream) {
  if (!a) return S2ST_ERR_ARG;
  return s2st_gemm(*a, (hipStream_t)stream);
}

int s2st_gemm_tile_f32(const s2st_gemm_args* a, int32_t* tile, void* stream) {
  if (!a || !tile) return S2ST_ERR_ARG;
  int t = 0;
  const int rc = s2st_gemm(*a, (hipStream_t)stream, &t);
  *tile = t;
  return rc;
}

int s2st_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int32_t rows, int32_t cols, float eps, void* stream) {
  return s2st_layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, cols, eps, (hipStream_t)stream);
}

int s2st_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx, int32_t dx_accumulate, float* dgamma, float* dbeta, float* scratch, int32_t rows, int32_t cols, void* stream) {
  return s2st_layernorm_bwd(dy, x, gamma, mean, rstd, dx, dx_accumulate, dgamma, dbeta, scratch, rows, cols, (hipStream_t)stream);
}

int s2st_softmax_fwd_f32(const float* s, float* p, float* pd, const int32_t* klen, int32_t B, int32_t H, int32_t T, int32_t S, int32_t ld, int32_t causal, float drop_p, uint64_t seed, void* stream) {
  return s2st_softmax_fwd(s, p, pd, klen, B, H, T, S, ld, causal, drop_p, seed, (hipStream_t)stream);
}

int s2st_softmax_bwd_f32(const float* p, const float* dpd, float* ds, int32_t B, int32_t H, int32_t T, int32_t S, int32_t ld, float drop_p, uint64_t seed, void* stream) {
  return s2st_softmax_bwd(p, dpd, ds, B, H, T, S, ld, drop_p, seed, (hipStream_t)stream);
}

int s2st_colsum_f32(const float* x, int64_t ld, int32_t rows, int32_t cols, float* out, int32_t accumulate, void* stream) {
  return s2st_colsum(x, ld, rows, cols, out, accumulate, (hipStream_t)stream);
}

int s2st_attn_headmean_f32(const float* p, float* out, int32_t B, int32_t H, int32_t T, int32_t S, int32_t ld, void* stream) {
  return s2st_attn_headmean(p, out, B, H, T, S, ld, (hipStream_t)stream);
}

int s2st_copy_rows_f32(const float* x, s2st_split xsp, float* y, s2st_split ysp, int32_t rows, int32_t C, void* stream) {
  return s2st_copy_rows(x, xsp, y, ysp, rows, C, (hipStream_t)stream);
}

int s2st_glu_fwd_f32(const float* a, float* y, s2st_split ysp, int32_t rows, int32_t C, void* stream) {
  return s2st_glu_fwd(a, y, ysp, rows, C, (hipStream_t)stream);
}

int s2st_glu_bwd_f32(const float* a, const float* dy, s2st_split dysp, float* da, s2st_split dasp, int32_t rows, int32_t C, void* stream) {
  return s2st_glu_bwd(a, dy, dysp, da, dasp, rows, C, (hipStream_t)stream);
}

int s2st_add_pe_f32(const float* x, float* y, const int32_t* pos, const float* table, int32_t rows, int32_t C, float scale, const float* alpha_ptr, float drop_p, uint64_t seed, void* stream) {
  return s2st_add_pe(x, y, pos, table, rows, C, scale, alpha_ptr, drop_p, seed, (hipStream_t)stream);
}

int s2st_pe_alpha_bwd_f32(const float* dy, const int32_t* pos, const float* table, int32_t rows, int32_t C, float drop_p, uint64_t seed, float* dalpha, void* stream) {
  return s2st_pe_alpha_bwd(dy, pos, table, rows, C, drop_p, seed, dalpha, (hipStream_t)stream);
}

int s2st_embed_fwd_f32(const int64_t* tokens, const float* table, float* y, int32_t rows, int32_t C, float scale, void* stream) {
  return s2st_embed_fwd((const long*)tokens, table, y, rows, C, scale, (hipStream_t)stream);
}

int s2st_embed_bwd_f32(const int64_t* tokens, const float* dy, float* dtable, int32_t rows, int32_t C, float scale, int64_t pad, void* stream) {
  return s2st_embed_bwd((const long*)tokens, dy, dtable, rows, C, scale, pad, (hipStream_t)stream);
}

int s2st_embed_bwd_ordered_f32(const int64_t* tokens, const float* dy, float* dtable, int32_t rows, int32_t C, int32_t V, float scale, int64_t pad, void* stream) {
  if (V <= 0) return S2ST_ERR_ARG;
  return s2st_embed_bwd((const long*)tokens, dy, dtable, rows, C, scale, (long)pad, (hipStream_t)stream, V);
}

int s2st_dropout_f32(const float* x, float* y, int64_t n, float a, float p, uint64_t seed, int32_t accumulate, void* stream) {
  return s2st_dropout(x, y, n, a, p, seed, accumulate, (hipStream_t)stream);
}

int s2st_relu_drop_bwd_f32(const float* dy, const float* y, float* dz, int64_t n, float p, void* stream) {
  return s2st_relu_drop_bwd(dy, y, dz, n, p, (hipStream_t)stream);
}

int s2st_conv_w_permute_f32(const float* w, float* wf, float* wd, int32_t O, int32_t I, int32_t Kw, void* stream) {
  return s2st_conv_w_permute(w, wf, wd, O, I, Kw, (hipStream_t)stream);
}

int s2st_conv_w_unpermute_acc_f32(const float* dwf, float* dw, int32_t O, int32_t I, int32_t Kw, void* stream) {
  return s2st_conv_w_unpermute_acc(dwf, dw, O, I, Kw, (hipStream_t)stream);
}

int s2st_bn_stats_f32(const float* x, int32_t rows, int32_t C, float* mean, float* var, float* run_mean, float* run_var, float momentum, float* tmp, void* stream) {
  return s2st_bn_stats(x, rows, C, mean, var, run_mean, run_var, momentum, tmp, (hipStream_t)stream);
}

int s2st_bn_apply_f32(const float* x, const float* mean, const float* var, const float* gamma, const float* beta, float* y, s2st_split ysp, const float* resid, int32_t rows, int32_t C, float eps, int32_t tanh_, float drop_p, uint64_t seed, void* stream) {
  return s2st_bn_apply(x, mean, var, gamma, beta, y, ysp, resid, rows, C, eps, tanh_, drop_p, seed, (hipStream_t)stream);
}

int s2st_bn_bwd_f32(const float* dy, s2st_split dysp, const float* x, const float* mean, const float* var, const float* gamma, const float* beta, float* dx, s2st_split dxsp, float* dgamma, float* dbeta, float* tmp, int32_t rows, int32_t C, float eps, int32_t tanh_, float drop_p, uint64_t seed, void* stream) {
  return s2st_bn_bwd(dy, dysp, x, mean, var, gamma, beta, dx, dxsp, dgamma, dbeta, tmp, rows, C, eps, tanh_, drop_p, seed, (hipStream_t)stream);
}

int s2st_mel_loss_f32(const float* feat, const float* post, const float* eos, const float* tgt, const int32_t* lens, int32_t B, int32_t D, int32_t F, float pos_weight, float* stats, float c_l1, float c_mse, float c_eos, float* dfeat, float* dpost, float* deos, void* stream) {
  return s2st_mel_loss(feat, post, eos, tgt, lens, B, D, F, pos_weight, stats, c_l1, c_mse, c_eos, dfeat, dpost, deos, (hipStream_t)stream);
}

int s2st_ls_ce_f32(const float* logits, const int64_t* target, int32_t rows, int32_t V, int64_t pad, float eps, float* stats, float* dlogits, float gscale, void* stream) {
  return s2st_ls_ce(logits, (const long*)target, rows, V, pad, eps, stats, dlogits, gscale, (hipStream_t)stream);
}

int s2st_ctc_f32(const float* logits, const int64_t* targets, int32_t Lmax, const int32_t* in_lens, const int32_t* tgt_lens, int32_t B, int32_t E, int32_t V, float* lprobs, float* loss_per_utt, float* dlogits, float gscale, float* ws, void* stream) {
  return s2st_ctc(logits, (const long*)targets, Lmax, in_lens, tgt_lens, B, E, V, lprobs, loss_per_utt, dlogits, gscale, ws, (hipStream_t)stream);
}

int s2st_sumsq_parts_f32(const float* x, int64_t n, float* parts, void* stream) {
  return s2st_sumsq_parts(x, n, parts, (hipStream_t)stream);
}
int64_t s2st_sumsq_parts_count(int64_t n) { return s2st_sumsq_nparts((long)n); }
int s2st_sumsq_f32(const float* x, int64_t n, float* out, void* stream) {
  return s2st_sumsq(x, n, out, (hipStream_t)stream);
}

int s2st_adam_f32(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq, float gmul, const float* gmul_dev, float max_norm, float lr, float beta1, float beta2, float eps, float wd, int32_t step, float* gnorm_out, void* p_bf16, int32_t* skipped, int32_t sumsq_parts, int32_t zero_grad, void* stream) {
  return s2st_adam(p, g, m, v, n, sumsq, gmul, gmul_dev, max_norm, lr, beta1, beta2, eps, wd, step, gnorm_out, (hipStream_t)stream, (uint16_t*)p_bf16, skipped, sumsq_parts, zero_grad);
}

int64_t s2st_layernorm_bwd_scratch(int32_t rows, int32_t cols) { return (int64_t)s2st_layernorm_bwd_blocks(rows, cols) * 2 * cols; }
int64_t s2st_ctc_workspace(int32_t B, int32_t E, int32_t Lmax) { return s2st_ctc_workspace_floats(B, E, Lmax); }

int s2st_flash_attn_fwd_bf16(const s2st_attn_args* args, void* stream) { return s2st_flash_attn_fwd(args, (hipStream_t)stream); }
int s2st_flash_attn_bwd_bf16(const s2st_attn_args* args, const float* dO, float* dvec_scratch, void* stream) {
  return s2st_flash_attn_bwd(args, dO, dvec_scratch, (hipStream_t)stream, 0);
}

int s2st_argmax_dim1_f32(const float* x, int64_t* idx, int32_t B, int32_t E, int32_t D, void* stream) {
  return s2st_argmax_dim1(x, (long*)idx, B, E, D, (hipStream_t)stream);
}
int s2st_affine_cols_f32(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int32_t C, void* stream) {
  return s2st_affine_cols(x, scale, shift, y, rows, C, (hipStream_t)stream);
}
int s2st_exp_transpose_f32(const float* x, float* y, int32_t T, int32_t C, void* stream) {
  return s2st_exp_transpose(x, y, T, C, (hipStream_t)stream);
}
int s2st_clamp_min_f32(float* x, int64_t n, float lo, void* stream) { return s2st_clamp_min(x, n, lo, (hipStream_t)stream); }
int s2st_gl_polar_f32(const float* mag, const float* ang, float* X, int32_t F, int32_t T, void* stream) {
  return s2st_gl_polar(mag, ang, X, F, T, (hipStream_t)stream);
}
int s2st_gl_project_f32(const float* mag, const float* Y, float* X, int32_t F, int32_t T, void* stream) {
  return s2st_gl_project(mag, Y, X, F, T, (hipStream_t)stream);
}
int s2st_reflect_pad_f32(const float* x, float* y, int32_t n, int32_t pad, void* stream) {
  return s2st_reflect_pad(x, y, n, pad, (hipStream_t)stream);
}
int s2st_gl_overlap_add_f32(const float* frames, const float* wsq, float* wave, int32_t T, int32_t n_fft, int32_t hop, int32_t n_out, void* stream) {
  return s2st_gl_overlap_add(frames, wsq, wave, T, n_fft, hop, n_out, (hipStream_t)stream);
}

int s2st_dtw_f32(const float* dist, const int32_t* shapes, int32_t B, int32_t M, int32_t N, float* cumdist, int32_t* backptr, int32_t* pathmap, void* stream) {
  return s2st_dtw(dist, shapes, B, M, N, cumdist, backptr, pathmap, (hipStream_t)stream);
}
int s2st_rms_dist_f32(const float* x1, const float* x2, float* out, int32_t m, int32_t n, int32_t D, int64_t ldo, void* stream) {
  return s2st_rms_dist(x1, x2, out, m, n, D, ldo, (hipStream_t)stream);
}
int s2st_power_spec_f32(const float* Y, float* P, int32_t T, int32_t F, void* stream) { return s2st_power_spec(Y, P, T, F, (hipStream_t)stream); }
int s2st_log_offset_f32(float* x, int64_t n, float eps, void* stream) { return s2st_log_offset(x, n, eps, (hipStream_t)stream); }

int s2st_gl_polar_split_f32(const float* mag, const float* aux, int32_t from_spectrum, const int32_t* tl, void* Xs, int32_t U, int32_t F, int32_t Fp, int32_t Tmax, void* stream) {
  return s2st_gl_polar_split(mag, aux, from_spectrum, tl, (uint16_t*)Xs, U, F, Fp, Tmax, (hipStream_t)stream);
}
int s2st_gl_frame_split_f32(const float* wave, const int32_t* tl, void* As, int32_t U, int32_t Tmax, int32_t hop, int32_t n_fft, int32_t Lw, void* stream) {
  return s2st_gl_frame_split(wave, tl, (uint16_t*)As, U, Tmax, hop, n_fft, Lw, (hipStream_t)stream);
}
int64_t s2st_hubert_conv0_stats_floats_i64(int32_t B, int32_t T, int32_t C) { return s2st_hubert_conv0_stats_floats(B, T, C); }
int s2st_hubert_conv0_gn_gelu_f32(const float* wave, const float* w, const float* gamma, const float* beta, float* y, uint16_t* y_bf16, float* stats, int32_t B, int32_t N, int32_t T, int32_t C, int32_t k, int32_t stride, float eps, void* stream) {
  return s2st_hubert_conv0_gn_gelu(wave, w, gamma, beta, y, y_bf16, stats, B, N, T, C, k, stride, eps, (hipStream_t)stream);
}
int s2st_decode_attn_f32(const float* q, int64_t ldq, void* k_cache, void* v_cache, int64_t ldk, int64_t kbs, const int32_t* klen, int32_t nkeys, int32_t B, int32_t H, int32_t dh, float scale, float* o, int64_t ldo, float* attn_mean, int32_t S, const float* k_new, const float* v_new, int64_t ld_new, int32_t pos_new, int32_t kv_bf16, void* stream) {
  return s2st_decode_attn(q, ldq, (float*)k_cache, (float*)v_cache, ldk, kbs, klen, nkeys, B, H, dh, scale, o, ldo, attn_mean, S, (hipStream_t)stream, k_new, v_new, ld_new, pos_new, kv_bf16);
}
int s2st_decode_stop_update_i32(const float* eos_prob, float thr, int32_t step, int32_t max_iter, int32_t B, int32_t* finished, int32_t* out_lens, int32_t* klen_next, int32_t* n_done, void* stream) {
  return s2st_decode_stop_update(eos_prob, thr, step, max_iter, B, finished, out_lens, klen_next, n_done, (hipStream_t)stream);
}
int s2st_gl_fft_supported_i32(int32_t n_fft) { return s2st_gl_fft_supported(n_fft) ? 1 : 0; }
int s2st_gl_polar_c_f32(const float* mag, const float* ang, const int32_t* tl, float* X, int32_t U, int32_t F, int32_t Tmax, void* stream) {
  return s2st_gl_polar_c(mag, ang, tl, X, U, F, Tmax, (hipStream_t)stream);
}
int s2st_exp_inplace_f32(float* x, int64_t n, void* stream) { return s2st_exp_inplace(x, n, (hipStream_t)stream); }
int s2st_gl_polar_u_f32(const float* mag, const double* uniform, const int64_t* offsets, const int32_t* tl, uint64_t seed, float* X, int32_t U, int32_t F, int32_t Tmax, void* stream) {
  return s2st_gl_polar_u(mag, uniform, (const long*)offsets, tl, seed, X, U, F, Tmax, (hipStream_t)stream);
}
int s2st_gl_stft_project_f32(const float* wave, const int32_t* tl, const float* win, const float* tw, const float* mag, float* X, int32_t U, int32_t Tmax, int32_t n_fft, int32_t hop, int32_t Lw, void* stream) {
  return s2st_gl_stft_project(wave, tl, win, tw, mag, X, U, Tmax, n_fft, hop, Lw, (hipStream_t)stream);
}
int s2st_gl_istft_frames_f32(const float* X, const int32_t* tl, const float* win, const float* tw, float* frames, int32_t U, int32_t Tmax, int32_t n_fft, int32_t hop, void* stream) {
  return s2st_gl_istft_frames(X, tl, win, tw, frames, U, Tmax, n_fft, hop, (hipStream_t)stream);
}
int s2st_gl_istft_ola_f32(const float* X, const int32_t* tl, const float* win, const float* tw, const float* wsq_all, const int64_t* wsq_off, float* wave, int32_t U, int32_t Tmax, int32_t n_fft, int32_t hop, int32_t Lw, void* stream) {
  return s2st_gl_istft_ola(X, tl, win, tw, wsq_all, (const long*)wsq_off, wave, U, Tmax, n_fft, hop, Lw, (hipStream_t)stream);
}
int s2st_gl_overlap_add_b_f32(const float* frames, const float* wsq_all, const int64_t* wsq_off, const int32_t* tl, float* wave, int32_t U, int32_t Tmax, int32_t n_fft, int32_t hop, int32_t Lw, void* stream) {
  return s2st_gl_overlap_add_b(frames, wsq_all, (const long*)wsq_off, tl, wave, U, Tmax, n_fft, hop, Lw, (hipStream_t)stream);
}

// Host-side integer work of the data path: the max-tokens batcher (the reference compiles it as a Cython
// extension, fairseq/data/data_utils_fast.pyx:20-100).  num_tokens[i] is the cost of the i-th index in the given
// order; batch_ends[k] receives the exclusive end position of batch k.  A running batch plus a tail: the tail
// joins the batch whenever the merged size is < bsz_mult or a multiple of it; on overflow of max_tokens
// (sentences x longest) or max_sentences the batch is closed and the tail starts the next one; a tail that
// overflows on its own is closed as well and the current item starts over.
int64_t s2st_batch_by_size(const int64_t* num_tokens, int64_t n, int64_t max_tokens, int64_t max_sentences,
                           int32_t bsz_mult, int32_t* batch_ends) {
  if (n < 0 || (n > 0 && (!num_tokens || !batch_ends)) || bsz_mult < 1 || n > 0x7fffffff) return S2ST_ERR_ARG;
  if (n == 0) return 0;
  if (max_tokens > 0)
    for (int64_t i = 0; i < n; ++i)
      if (num_tokens[i] > max_tokens) return S2ST_ERR_SHAPE;  // "Sentences lengths should not exceed max_tokens"
  for (int64_t i = 0; i <= n; ++i) batch_ends[i] = 0;  // n + 1 slots
  int64_t count = 0, batch_start = 0, tail_max = 0, batch_max = 0;
  for (int64_t pos = 0; pos < n; ++pos) {
    tail_max = tail_max > num_tokens[pos] ? tail_max : num_tokens[pos];
    const int64_t new_end = pos + 1;
    int64_t new_max = batch_max > tail_max ? batch_max : tail_max;
    const int64_t sentences = new_end - batch_start;
    const bool overflow = (max_sentences > 0 && sentences > max_sentences) || (max_tokens > 0 && sentences * new_max > max_tokens);
    const bool fits_mult = sentences < bsz_mult || sentences % bsz_mult == 0;
    if (overflow) {
      const int64_t tail_tokens = tail_max * (new_end - batch_ends[count]);
      if (max_tokens > 0 && tail_tokens > max_tokens) {
        batch_ends[++count] = (int32_t)pos;
        tail_max = num_tokens[pos];
      }
      batch_start = batch_ends[count];
      ++count;
      new_max = tail_max;
    }
    if (overflow || fits_mult) {
      batch_ends[count] = (int32_t)new_end;
      batch_max = new_max;
      tail_max = 0;
    }
  }
  if (batch_ends[count] != n) ++count;
  return count;
}

int s2st_gemm_skinny_f32(const float* x, int64_t ldx, const void* w_bf16, int64_t ldw, float* y, int64_t ldy, const float* bias, int32_t act, float drop_p, uint64_t seed, const float* resid, int64_t ldr, int32_t M, int32_t N, int32_t K, void* stream) {
  return s2st_gemm_skinny(x, ldx, (const bf16raw*)w_bf16, ldw, y, ldy, bias, act, drop_p, seed, resid, ldr, M, N, K, (hipStream_t)stream);
}
int s2st_ln_gemm_skinny_f32(const float* x, int64_t ldx, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w_bf16, int64_t ldw, float* y, int64_t ldy, const float* bias, int32_t act, int32_t M, int32_t N, int32_t K, void* stream) {
  if (!ln_gamma || !ln_beta) return S2ST_ERR_ARG;
  return s2st_gemm_skinny(x, ldx, (const bf16raw*)w_bf16, ldw, y, ldy, bias, act, 0.f, 0, nullptr, 0, M, N, K, (hipStream_t)stream, ln_gamma, ln_beta, ln_eps);
}

int s2st_gemm_group_f32(const s2st_gemm_args* list, int32_t n, void* stream) {
  if (!list || n < 0) return S2ST_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (!s2st_gemm_group_ok(list[i])) return S2ST_ERR_SHAPE;
  return s2st_gemm_bf16_group(list, n, (hipStream_t)stream);
}
int s2st_log_softmax_rows_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t rows, int32_t V, int32_t log_out, void* stream) {
  return s2st_log_softmax_rows(x, ldx, y, ldy, rows, V, log_out, (hipStream_t)stream);
}
int64_t s2st_gemm_streamk_scratch_floats(void) { return S2ST_STREAMK_SCRATCH_FLOATS; }
int s2st_gemm_streamk_scratch(float* scratch, int64_t floats, void* stream) {
  if (scratch && floats < S2ST_STREAMK_SCRATCH_FLOATS) return S2ST_ERR_WORKSPACE;
  s2st_gemm_streamk_bind((hipStream_t)stream, scratch, floats);
  return 0;
}
#ifndef S2ST_SOURCE_HASH
#define S2ST_SOURCE_HASH "unknown"  // (__graft_entry__.build passes the hash of the sources; the test emulator build does not)
#endif
int s2st_source_hash(char* out, int32_t cap) {
  const char* h = S2ST_SOURCE_HASH;
  if (!out || cap < 2) return S2ST_ERR_ARG;
  int i = 0;
  for (; h[i] && i < cap - 1; ++i) out[i] = h[i];
  out[i] = 0;
  return 0;
}
int s2st_profile_enable(int32_t enable) { s2st_profile_enable_impl(enable); return 0; }
int64_t s2st_profile_report(char* out, int64_t cap) { return s2st_profile_report_impl(out, cap, 0); }
int64_t s2st_profile_timeline(char* out, int64_t cap) { return s2st_profile_report_impl(out, cap, 1); }

}  // extern "C"
