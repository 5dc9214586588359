/* C ABI of libs2st_hip.so -- the MI355X-native (gfx950) kernels and training engine behind
 * the s2st_transformer hot path.
 *
 * The reference has no FFI on this path: its operators are ATen calls made from Python
 * (SURVEY.md section 8(b)).  Each entry point below therefore cites the reference call
 * site(s) whose arithmetic it replaces.  Conventions: plain pointers + sizes, caller-owned
 * DEVICE buffers, stream-ordered (`stream` is a hipStream_t passed as void*; NULL = the
 * default stream), re-entrant, return 0 on success or a negative code (no exceptions cross
 * the boundary).  All floating-point buffers are fp32; token ids are int64, lengths int32.
 */
#ifndef S2ST_HIP_H
#define S2ST_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define S2ST_OK 0
#define S2ST_ERR_LAUNCH (-1)
#define S2ST_ERR_SHAPE (-2)
#define S2ST_ERR_WORKSPACE (-3)
#define S2ST_ERR_ARG (-4)
#define S2ST_ERR_COMM (-5) /* no RCCL library could be bound, or an RCCL call failed (message on stderr) */

/* offset of "slow index" i:  per <= 0 ? i*ld : (i / per) * bs + (i % per) * ld          */
typedef struct { int64_t ld; int64_t bs; int32_t per; int32_t _pad; } s2st_split;

typedef struct {
  const void* p;  /* fp32 (dtype 0) or bf16 (dtype 1, raw uint16) elements */
  int32_t kmajor; /* 1: X(r,k) at p + split(r) + k ; 0: X(r,k) at p + split(k) + r */
  int32_t dtype;  /* S2ST_F32 / S2ST_BF16; all strides below are in ELEMENTS of this type */
  s2st_split sp;
  int64_t zo, zi; /* batch strides: z -> (z / zdiv) * zo + (z % zdiv) * zi */
} s2st_gemm_operand;

#define S2ST_F32 0
#define S2ST_BF16 1

typedef struct {
  float* p;      /* fp32 result, or NULL when only the bf16 copy is wanted */
  s2st_split sp; /* C(m,n) at p + split(m) + n */
  int64_t zo, zi;
  uint16_t* h;   /* optional bf16 (round-to-nearest-even) copy of the result, addressed like p */
} s2st_gemm_out;

typedef struct {
  float alpha;
  int32_t act;        /* 0 none, 1 relu, 2 exact (erf) gelu */
  const float* bias;  /* [N] or NULL */
  float drop_p;       /* dropout after activation, 0 = off */
  int32_t accumulate; /* C += value */
  uint64_t seed;
  const float* resid; /* added last; addressed like C; or NULL */
  /* backward-of-activation epilogue (bf16-operand path): the result is the gradient w.r.t. the OUTPUT
   * y of a previous ReLU+dropout layer; value = y(m,n) != 0 ? value * mask_scale : 0 (y given as its
   * bf16 copy, addressed like C), and colsum[n] += sum_m value (that layer's bias gradient). */
  const uint16_t* mask_y;
  float mask_scale;
  float* colsum;
  /* colsum_part != NULL: instead of atomic adds into colsum, every (row tile, wave row) writes its partial column sums
   * to colsum_part[(2 * tile_m + wave_m) * N + n] -- 2 * ceil(M / tile rows) partial rows (the tile rows come back from
   * s2st_gemm_f32_tile), folded in index order by the caller: run-to-run identical sums */
  float* colsum_part;
  /* batched products: the bias of batch z is bias + (z / zdiv) * bias_zo (0: one bias for all) */
  int64_t bias_zo;
} s2st_gemm_epilogue;

typedef struct {
  s2st_gemm_operand A, B;
  s2st_gemm_out C;
  s2st_gemm_epilogue ep;
  int32_t M, N, K;
  int32_t batch, zdiv;
  int32_t precise; /* 0: bf16 MFMA; 1: bf16x3 split (~fp32 accuracy, parity tests; fp32 operands only) */
  int32_t splitk, kchunk, avec, bvec, cvec, tiles_n; /* filled by the launcher */
  /* optional caller-owned scratch for split-K partial sums (bf16-operand path): accumulating
   * GEMMs with few output tiles (weight gradients) split K over workgroups, write fp32 partial
   * slabs here and combine them with a second kernel.  NULL: fp32 atomics into C instead. */
  float* ws;
  int64_t ws_floats;
  float* slab; /* filled by the launcher */
} s2st_gemm_args;

/* C(m,n) = epi(alpha * sum_k A(m,k) B(n,k)).  Operands are both fp32 (converted to bf16 on the
 * way into LDS) or both bf16 (the fast path: producers keep bf16 copies of every GEMM operand so
 * the rounding is the same as the fp32 path's, at half the bytes).  bf16 operands must have
 * 16-byte aligned bases, ld % 8 == 0 and rows readable up to the next multiple of 8 elements of
 * their contiguous extent (padding content is ignored); otherwise a guarded scalar path runs.
 * Replaces F.linear / F.conv1d / torch.bmm at
 * fairseq/modules/multihead_attention.py:170-192,332,367, transformer_layer.py:158-162,
 * examples/s2s_trans/models/s2st_transformer.py:135-139,452-455, tacotron2.py:95-126. */
int s2st_gemm_f32(const s2st_gemm_args* args, void* stream);
/* the same launch; *tile = 1000 * tile rows + tile columns of the kernel form the launcher picked for a bf16 product (0 for
 * the fp32-operand path): lets a caller -- the tests, a tuning tool -- see which form ran (128 x 128 / 128 x 64 / 64 x 64
 * ring or 4-wave forms, the 256 x 256 four-phase form of gemm_bf16_p4.hip) */
int s2st_gemm_tile_f32(const s2st_gemm_args* args, int32_t* tile, void* stream);

/* Stream-K for the bf16 products launched on `stream` (s2st_gemm_f32 / s2st_gemm_group_f32): with a scratch buffer of
 * s2st_gemm_streamk_scratch_floats() floats bound to the stream (first 4 KiB zero at bind time; NULL unbinds), launches
 * whose 128 x 128 tiles do not fill whole rounds of CUs share the K-steps of all tiles evenly between one workgroup per
 * CU; partial accumulators meet in the scratch.  Results equal the unsplit launch up to fp32 summation order. */
int64_t s2st_gemm_streamk_scratch_floats(void);
int s2st_gemm_streamk_scratch(float* scratch, int64_t floats, void* stream);

/* Up to 8 bf16-operand products (same operand layouts, batch 1, M, N >= 128, 16-byte aligned operands) in ONE persistent
 * launch: a layer's weight-gradient GEMMs dW = dY^T X (torch.mm calls inside autograd's backward of every F.linear,
 * fairseq/modules/transformer_layer.py:140-162) with K = tokens unsplit.  S2ST_ERR_SHAPE if a problem does not qualify. */
int s2st_gemm_group_f32(const s2st_gemm_args* list, int32_t n, void* stream);

/* Fused multi-head attention (bf16 operands, head width 64 or 128): masks + fp32 online softmax +
 * dropout + P*V in one kernel; backward recomputes the probabilities from the saved log-sum-exp.
 * Replaces fairseq/modules/multihead_attention.py:224-367 (and the same steps inside
 * F.multi_head_attention_forward, :170-192).  Row (b, t) of head h of q lives at
 * q + (b*T + t)*ldq + h*dh (bf16 elements); k / v likewise with S rows; o, oh and doh are dense
 * [B*T][H*dh]; dq / dk / dv (fp32) are addressed like q / k / v.  The dropout mask of element
 * (b, h, t, s) is drop(seed, ((b*H + h)*T + t)*ld_drop + s) -- the unfused kernels' index space. */
typedef struct {
  const uint16_t *q, *k, *v;
  int64_t ldq, ldk, ldv;
  float* o;
  uint16_t* oh;       /* optional bf16 copy of o */
  float* lse;         /* [B*H*T] log-sum-exp of the masked, scaled scores */
  const int32_t* klen; /* [B] valid keys per batch element, or NULL */
  int32_t B, H, T, S, dh, causal;
  float scale, drop_p;
  uint64_t seed;
  int32_t ld_drop;
  const uint16_t* doh; /* backward: bf16 dO */
  float *dq, *dk, *dv; /* backward outputs (overwritten); each may be NULL when its bf16 twin is given */
  uint16_t *dqh, *dkh, *dvh; /* optional bf16 gradients, addressed like q / k / v */
  float *dbq, *dbk, *dbv;    /* optional: += column sums of dq / dk / dv (the projections' bias gradients, [H*dh]) */
} s2st_attn_args;
int s2st_flash_attn_fwd_bf16(const s2st_attn_args* args, void* stream);
/* dO: fp32 [B*T][H*dh]; dvec_scratch: B*H*T floats */
int s2st_flash_attn_bwd_bf16(const s2st_attn_args* args, const float* dO, float* dvec_scratch, void* stream);

/* fairseq/modules/layer_norm.py:11-35 (LayerNorm / FusedLayerNorm), forward; saves mean, rstd */
int s2st_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int32_t rows, int32_t cols, float eps, void* stream);

/* backward of the above: dx (=|+=), dgamma += , dbeta += ; scratch: s2st_layernorm_bwd_scratch floats */
int s2st_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx, int32_t dx_accumulate, float* dgamma, float* dbeta, float* scratch, int32_t rows, int32_t cols, void* stream);
int64_t s2st_layernorm_bwd_scratch(int32_t rows, int32_t cols);

/* multihead_attention.py:343-366: key-padding / causal -inf masks, fp32 softmax, dropout. scores [B,H,T,ld] */
int s2st_softmax_fwd_f32(const float* s, float* p, float* pd, const int32_t* klen, int32_t B, int32_t H, int32_t T, int32_t S, int32_t ld, int32_t causal, float drop_p, uint64_t seed, void* stream);

/* ds = p * (dp' - sum dp' p), dp' = dropmask * dpd */
int s2st_softmax_bwd_f32(const float* p, const float* dpd, float* ds, int32_t B, int32_t H, int32_t T, int32_t S, int32_t ld, float drop_p, uint64_t seed, void* stream);

/* bias gradients: out[c] (+)= sum_r x[r][c] */
int s2st_colsum_f32(const float* x, int64_t ld, int32_t rows, int32_t cols, float* out, int32_t accumulate, void* stream);

/* s2st_transformer.py:424-427: mean over heads, [B,H,T,S] -> [B,S,T] */
int s2st_attn_headmean_f32(const float* p, float* out, int32_t B, int32_t H, int32_t T, int32_t S, int32_t ld, void* stream);

/* row copy between split-addressed buffers (halo padding for the convs) */
int s2st_copy_rows_f32(const float* x, s2st_split xsp, float* y, s2st_split ysp, int32_t rows, int32_t C, void* stream);

/* s2st_transformer.py:137 F.glu(dim=channels): a [rows][2C] -> y rows via split */
int s2st_glu_fwd_f32(const float* a, float* y, s2st_split ysp, int32_t rows, int32_t C, void* stream);

/* backward of GLU */
int s2st_glu_bwd_f32(const float* a, const float* dy, s2st_split dysp, float* da, s2st_split dasp, int32_t rows, int32_t C, void* stream);

/* s2st_transformer.py:197-208, 385-387: y = dropout(scale*x + alpha*PE[pos]) */
int s2st_add_pe_f32(const float* x, float* y, const int32_t* pos, const float* table, int32_t rows, int32_t C, float scale, const float* alpha_ptr, float drop_p, uint64_t seed, void* stream);

/* gradient of decoder.pos_emb_alpha */
int s2st_pe_alpha_bwd_f32(const float* dy, const int32_t* pos, const float* table, int32_t rows, int32_t C, float drop_p, uint64_t seed, float* dalpha, void* stream);

/* transformer_decoder.py:303: embed_scale * embed_tokens(tokens) */
int s2st_embed_fwd_f32(const int64_t* tokens, const float* table, float* y, int32_t rows, int32_t C, float scale, void* stream);

/* embedding gradient (pad row receives none) */
int s2st_embed_bwd_f32(const int64_t* tokens, const float* dy, float* dtable, int32_t rows, int32_t C, float scale, int64_t pad, void* stream);
/* the same gradient with the rows of every token id added in row order (no atomics: repeats bit for bit; what the engine
 * uses); V = number of ids, C <= 1024 */
int s2st_embed_bwd_ordered_f32(const int64_t* tokens, const float* dy, float* dtable, int32_t rows, int32_t C, int32_t V, float scale, int64_t pad, void* stream);

/* fairseq_dropout.py:16-27: y (+)= a * x * mask(seed) */
int s2st_dropout_f32(const float* x, float* y, int64_t n, float a, float p, uint64_t seed, int32_t accumulate, void* stream);

/* backward of dropout(relu(z)) from its output */
int s2st_relu_drop_bwd_f32(const float* dy, const float* y, float* dz, int64_t n, float p, void* stream);

/* Conv1d weight [O][I][Kw] -> GEMM layouts [O][Kw][I] and flipped [I][Kw][O] */
int s2st_conv_w_permute_f32(const float* w, float* wf, float* wd, int32_t O, int32_t I, int32_t Kw, void* stream);

/* dW[O][I][Kw] += dWf[O][Kw][I] */
int s2st_conv_w_unpermute_acc_f32(const float* dwf, float* dw, int32_t O, int32_t I, int32_t Kw, void* stream);

/* tacotron2.py:112,125 BatchNorm1d(train): batch mean/var over all rows + running stats; tmp: S2ST_BN_TMP_FLOATS(C) floats.
 * Scratch of the BatchNorm statistics / backward reductions: results (2 C) and slab partials (64 x 2 C) -- the
 * reductions are fixed-order (no atomics), so the statistics and everything downstream are reproducible */
#define S2ST_BN_TMP_FLOATS(C) (130 * (long)(C))
int s2st_bn_stats_f32(const float* x, int32_t rows, int32_t C, float* mean, float* var, float* run_mean, float* run_var, float momentum, float* tmp, void* stream);

/* y = dropout([tanh](gamma*xhat+beta)) (+resid) */
int s2st_bn_apply_f32(const float* x, const float* mean, const float* var, const float* gamma, const float* beta, float* y, s2st_split ysp, const float* resid, int32_t rows, int32_t C, float eps, int32_t tanh_, float drop_p, uint64_t seed, void* stream);

/* backward of bn_apply (train-mode statistics); tmp: S2ST_BN_TMP_FLOATS(C) floats */
int s2st_bn_bwd_f32(const float* dy, s2st_split dysp, const float* x, const float* mean, const float* var, const float* gamma, const float* beta, float* dx, s2st_split dxsp, float* dgamma, float* dbeta, float* tmp, int32_t rows, int32_t C, float eps, int32_t tanh_, float drop_p, uint64_t seed, void* stream);

/* s2st_loss.py:294-315 compute_loss: masked L1+MSE (pre/post-net) + BCE(pos_weight) sums and gradients */
int s2st_mel_loss_f32(const float* feat, const float* post, const float* eos, const float* tgt, const int32_t* lens, int32_t B, int32_t D, int32_t F, float pos_weight, float* stats, float c_l1, float c_mse, float c_eos, float* dfeat, float* dpost, float* deos, void* stream);

/* s2st_loss.py:33-50, 330-348: log_softmax + label-smoothed NLL + accuracy, and d/dlogits */
int s2st_ls_ce_f32(const float* logits, const int64_t* target, int32_t rows, int32_t V, int64_t pad, float eps, float* stats, float* dlogits, float gscale, void* stream);

/* s2st_transformer.py:458-463 get_normalized_probs (fairseq/utils.py log_softmax / softmax, fp32): y[r][0..V) =
 * log_out ? log_softmax(x[r]) : softmax(x[r]); row strides ldx / ldy in floats */
int s2st_log_softmax_rows_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t rows, int32_t V, int32_t log_out, void* stream);

/* s2st_loss.py:229-243 + s2st_transformer.py:458-463: log_softmax + CTCLoss(mean, zero_infinity); ws from s2st_ctc_workspace_floats */
int s2st_ctc_f32(const float* logits, const int64_t* targets, int32_t Lmax, const int32_t* in_lens, const int32_t* tgt_lens, int32_t B, int32_t E, int32_t V, float* lprobs, float* loss_per_utt, float* dlogits, float gscale, float* ws, void* stream);

/* fairseq/utils.py:345-385 gradient L2 norm (sum of squares, out +=) */
int s2st_sumsq_f32(const float* x, int64_t n, float* out, void* stream);
/* the same sum left as per-block partial sums in parts[0 .. s2st_sumsq_parts_count(n)) (count <= 1024): plain stores,
 * nothing to zero beforehand; s2st_adam_f32(sumsq = parts, sumsq_parts = count) adds them in index order, so the norm --
 * and everything the clip coefficient touches -- repeats bit for bit from run to run */
int s2st_sumsq_parts_f32(const float* x, int64_t n, float* parts, void* stream);
int64_t s2st_sumsq_parts_count(int64_t n);

/* trainer.py:838-873 + adam.py:163-239: grad scale (gmul * *gmul_dev), clip-by-norm, fairseq Adam on a flat arena.
 * p_bf16 (optional): bf16 copy of the updated parameters (the GEMM-operand copy of the next forward, see
 * s2st_engine_bind_bf16 / s2st_engine_bf16_is_fresh).  skipped (optional, device int32): incremented when the
 * gradient norm is non-finite; parameters, moments and gradients are then left untouched (the reference raises
 * FloatingPointError at trainer.py:860-867 -- the host reads this counter at its logging interval and does the same).
 * sumsq_parts: 0 = sumsq is one float; > 0 = sumsq holds that many partial sums (s2st_sumsq_parts_f32).
 * zero_grad: 0 = g holds the scaled and clipped gradient afterwards; 1 = g is left all zero, also when the update is
 * skipped (the write the kernel makes anyway: the next step's zero_grad() pass over the arena is not needed) */
int s2st_adam_f32(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq, float gmul, const float* gmul_dev, float max_norm, float lr, float beta1, float beta2, float eps, float wd, int32_t step, float* gnorm_out, void* p_bf16, int32_t* skipped, int32_t sumsq_parts, int32_t zero_grad, void* stream);

/* --grad-exchange-dtype bf16 (runtime/distributed.py; opt-in, the default exchange is fp32 like the reference's DDP,
 * fairseq/models/distributed_fairseq_model.py:58-67): a finished range of the gradient arena rounded to bf16 (RNE) for
 * the collective, and the reduced values widened back into the fp32 arena.  g 16-byte aligned, the bf16 buffer 8-byte. */
int s2st_grad_pack_bf16_f32(const float* g, uint16_t* out, int64_t n, void* stream);
/* One-GPU stand-in for the kernels of the gradient all-reduce (fairseq/models/distributed_fairseq_model.py:58-67 wraps the
 * model in torch DDP, whose reducer all-reduces 25 MB buckets over NCCL beside the backward): `wgs` workgroups read move_bytes
 * from bucket [n floats] (cyclically) and write them to scratch [n floats], paced to gbps GB/s.  Changes no gradient; lets a
 * single GPU measure what a bandwidth-moving neighbour on a few CUs costs the training step (bench.py --exchange-proxy). */
int s2st_exchange_proxy_f32(const float* bucket, float* scratch, int64_t n, int64_t move_bytes, int32_t wgs, float gbps, void* stream);
int s2st_grad_unpack_bf16_f32(const uint16_t* in, float* g, int64_t n, void* stream);

/* floats of workspace s2st_ctc_f32 needs */
int64_t s2st_ctc_workspace(int32_t B, int32_t E, int32_t Lmax);

/* ======================================================================================
 * Training engine: the whole s2st_transformer forward (+ s2st_loss) and backward as one
 * stream-ordered schedule of the kernels above, over flat parameter / gradient arenas.
 * Replaces, for the hot path, S2STTransformerModel.forward
 * (examples/s2s_trans/models/s2st_transformer.py:752-786, with encoder :195-237, decoder
 * :369-456, aux decoders :483-578 over fairseq/models/transformer/transformer_decoder.py:253-378)
 * and Tacotron2Criterion.forward (criterions/s2st_loss.py:179-292) plus autograd's backward.
 * ====================================================================================== */
typedef struct {
  int32_t enc_layers, dec_layers, enc_dim, dec_dim, enc_ffn, dec_ffn, enc_heads, dec_heads;
  int32_t enc_pre_ln, dec_pre_ln;
  int32_t in_dim, conv_channels, conv_k; /* two conv layers, stride 2 (conv_kernel_sizes "k,k") */
  int32_t out_dim;                        /* output_frame_dim * n_frames_per_step */
  int32_t prenet_layers, prenet_dim, postnet_layers, postnet_dim, postnet_k;
  int32_t tap_asr, tap_st;                /* --middle-layers ids (0-based, after the layer); -1 none */
  int32_t has_asr, has_st, has_ctc;
  int32_t asr_layers, asr_dim, st_layers, st_dim, src_vocab, tgt_vocab;
  int32_t no_scale_embedding;
  int32_t precise;                        /* 1: bf16x3 GEMMs (parity mode) */
  /* s2st_transformer_mtl (examples/s2s_trans/models/s2st_transformer_mtl.py:223-386): a second CTC head, over the TARGET
   * text, on the raw output of decoder layer tap_dec (--middle-layers-decoder); -1 / 0: none */
  int32_t tap_dec, has_ctc_tgt;
  /* t2s_transformer (examples/s2s_trans/models/t2s_transformer.py:37-126): a TEXT encoder front in place of the speech
   * subsampler -- token embedding, enc_conv_layers x (Conv1d k=enc_conv_k + BatchNorm1d + ReLU + dropout), a linear
   * projection, alpha-scaled positions; the batch's src_txt / src_txt_lens are then the encoder input (S = E = Ls) */
  int32_t text_input, enc_conv_layers, enc_conv_k;
  /* speaker conditioning (examples/s2s_trans/models/s2st_transformer.py:203-206, 441-444; tables from
   * tasks/s2s_translation.py:145-170): n_speakers > 0 adds encoder.embed_speaker.weight [n_speakers, enc_dim] -- the
   * utterance's row is added to every encoder position before the dropout -- and decoder.embed_speaker.weight
   * [n_speakers, out_dim] -- the row replaces the first frame of prev_output_tokens.  spk_frozen: tables loaded with
   * Embedding.from_pretrained(freeze=True): no gradient is formed for them. */
  int32_t n_speakers, spk_frozen;
  /* t2s_transformer with speakers (examples/s2s_trans/models/t2s_transformer.py:43-46, 107-111): the table is
   * encoder.embed_speaker.weight [n_speakers, spk_dim] (--speaker-embed-dim) and the encoder output becomes
   * encoder.spk_emb_proj(cat[x, row]) with weight [enc_dim, enc_dim + spk_dim]; unused (0) for speech input, where the
   * rows are added and their widths are fixed by enc_dim / out_dim.  Frozen tables (spk_frozen) are reported as
   * BUFFERS by s2st_engine_param_info: they live next to the BatchNorm statistics, outside the optimizer's arena. */
  int32_t spk_dim;
  float dropout, attn_dropout, act_dropout, prenet_dropout, postnet_dropout;
  float ctc_weight, asr_weight, st_weight, w_l1, w_mse, w_eos, bce_pos_weight, label_smoothing;
  float ctc_tgt_weight;                   /* s2st_loss_mtl.py:171-185 */
  float enc_dropout;                      /* t2s encoder prenet dropout (nn.Dropout: training only) */
  /* s2t_transformer_hubert (fairseq/models/speech_to_text/s2t_transformer_me.py:82-330): the ST / ASR pre-training stage of
   * the mix- / prompt-tuning recipes (run_mix_tuning.sh:100) -- the SAME speech encoder, and in place of the mel decoder
   * ONE full-width text decoder (fairseq TransformerDecoder: dec_layers x dec_dim, embedding and output projection over
   * tgt_vocab, parameter names decoder.embed_tokens / decoder.layers.N / decoder.layer_norm / decoder.output_projection)
   * reading the encoder output.  The batch's prev_src_txt / src_txt / src_txt_lens / src_txt_pos / pe_asr (width dec_dim)
   * slots carry the decoder's tokens -- the host picks source or target text by the criterion's --test-type
   * (criterions/s2t_loss.py:88-92) -- and the loss is s2t_loss.py:36-56's label-smoothed NLL SUMMED over the non-pad
   * tokens: stats[S2ST_STAT_LOSS] = that sum, S2ST_STAT_ASR_NLL / _SMOOTH / _CORRECT / _TOTAL its parts and the accuracy
   * counts; outputs.asr_logits = [B, Ls, tgt_vocab].  No mel decoder, post-net, CTC or aux heads exist in this mode. */
  int32_t s2t_mode;
} s2st_model_config;

typedef struct {
  char name[120];
  int64_t offset; /* element offset in the flat arena (params: fp32 trainable; buffers: BN stats) */
  int64_t numel;
  int32_t ndim;
  int32_t shape[4];
  int32_t is_buffer; /* 0: parameter arena, 1: buffer arena */
} s2st_param_info;

typedef struct {
  int32_t B, S, D, Ls, Lt;        /* batch, max src frames, max decoder steps, max text lens */
  int32_t E;                      /* encoder positions = conv-subsampled S (host computes) */
  const float* src;               /* [B,S,in_dim] */
  const int32_t* enc_lens;        /* [B] valid encoder positions */
  const int32_t* enc_pos;         /* [B*E] positional-table rows (make_positions on the pad mask) */
  const int32_t* ctc_in_lens;     /* [B] (s2st_loss.py:231-232) */
  const float* prev;              /* [B,D,out_dim] prev_output_tokens */
  const float* tgt;               /* [B,D,out_dim] tgt_speech */
  const int32_t* tgt_lens;        /* [B] */
  const int32_t* dec_pos;         /* [B*D] */
  const int64_t* prev_src_txt;    /* [B,Ls] */
  const int64_t* src_txt;         /* [B,Ls] */
  const int32_t* src_txt_lens;    /* [B] */
  const int32_t* src_txt_pos;     /* [B*Ls] */
  const int64_t* prev_tgt_txt;    /* [B,Lt] */
  const int64_t* tgt_txt;         /* [B,Lt] */
  const int32_t* tgt_txt_lens;    /* [B] */
  const int32_t* tgt_txt_pos;     /* [B*Lt] */
  /* sinusoidal tables [>= T+2][dim] (row 1 = padding row = zeros), built once by the host */
  const float* pe_enc;            /* dim enc_dim, rows >= E + 2 */
  const float* pe_dec;            /* dim dec_dim, rows >= D + 2 */
  const float* pe_asr;            /* dim asr_dim, rows >= Ls + 2 */
  const float* pe_st;             /* dim st_dim,  rows >= Lt + 2 */
  int32_t ntokens, src_txt_ntokens, tgt_txt_ntokens; /* host-side sums */
  int32_t training;               /* BatchNorm batch stats + dropout */
  int32_t want_attn;              /* also produce the head-averaged alignment [B,E,D] */
  uint64_t seed;                  /* dropout seed of this step */
  const int64_t* speaker;         /* [B] speaker ids (sample["speaker"]); NULL = unconditioned */
} s2st_batch;

typedef struct { /* caller-owned device outputs; any pointer may be NULL (kept internal) */
  float* post_feat;   /* [B,D,out_dim] */
  float* feat;        /* [B,D,out_dim] feature_out */
  float* eos;         /* [B,D] */
  float* attn;        /* [B,E,D] */
  float* enc_out;     /* [B,E,C] (reference layout is [E,B,C]: transpose the view) */
  float* tap0;        /* [B,E,C] */
  float* tap1;        /* [B,E,C] */
  float* asr_logits;  /* [B,Ls,src_vocab] */
  float* st_logits;   /* [B,Lt,tgt_vocab] */
  float* ctc_lprobs;  /* [B,E,src_vocab] */
  float* stats;       /* [32]: see S2ST_STAT_* */
} s2st_outputs;

enum {
  S2ST_STAT_L1_SUM = 0, S2ST_STAT_MSE_SUM = 1, S2ST_STAT_BCE_SUM = 2,
  S2ST_STAT_ASR_NLL = 3, S2ST_STAT_ASR_SMOOTH = 4, S2ST_STAT_ASR_CORRECT = 5, S2ST_STAT_ASR_TOTAL = 6,
  S2ST_STAT_ST_NLL = 7, S2ST_STAT_ST_SMOOTH = 8, S2ST_STAT_ST_CORRECT = 9, S2ST_STAT_ST_TOTAL = 10,
  S2ST_STAT_LOSS = 16, S2ST_STAT_L1 = 17, S2ST_STAT_MSE = 18, S2ST_STAT_EOS = 19,
  S2ST_STAT_CTC = 20, S2ST_STAT_ASR = 21, S2ST_STAT_ST = 22, S2ST_STAT_CTC_TGT = 23, S2ST_STAT_GNORM = 24
};

typedef struct s2st_engine s2st_engine;

int s2st_engine_create(const s2st_model_config* cfg, s2st_engine** out);
void s2st_engine_destroy(s2st_engine* e);
int32_t s2st_engine_num_params(const s2st_engine* e);
int s2st_engine_param_info(const s2st_engine* e, int32_t i, s2st_param_info* out);
int64_t s2st_engine_param_floats(const s2st_engine* e);  /* size of the parameter arena */
int64_t s2st_engine_buffer_floats(const s2st_engine* e); /* size of the buffer arena */
/* bind caller-owned device arenas: params/grads [param_floats], buffers [buffer_floats] */
int s2st_engine_bind(s2st_engine* e, float* params, float* grads, float* buffers);
/* fast (precise == 0) mode only: caller-owned bf16 arena of param_floats elements; the engine
 * refreshes it from `params` at the start of every forward and feeds the GEMMs from it */
int s2st_engine_bind_bf16(s2st_engine* e, uint16_t* params_bf16);
/* The caller states that the bf16 arena already equals bf16(params) (s2st_adam_f32 wrote it with the update):
 * the NEXT forward skips its refresh pass.  One-shot: any later forward refreshes again unless told anew. */
int s2st_engine_bf16_is_fresh(s2st_engine* e);
/* instrumentation (S2ST_STALL_TRACE=1): time the data-path stream spent in cross-stream waits since the last
 * report, in microseconds (per-wait lines on stderr when verbose); call after synchronising the device */
int64_t s2st_engine_stall_report(s2st_engine* e, int32_t verbose);
/* Dropout sites of a forward (test instrumentation: parity with the recipe's dropouts ON).  The reference draws its masks
 * from torch's generator at every F.dropout (fairseq/modules/fairseq_dropout.py:16-27; sites: transformer_layer.py:150-162,
 * 384-431; multihead_attention.py:360-366; s2st_transformer.py:197-208, 385-388; tacotron2.py:95-98, 122-126;
 * transformer_decoder.py:281-370); this library keeps no masks -- keep(element) = hash(site seed, element index) -- so the
 * only way to compare a dropout-on step with the CPU oracle is to give the oracle THIS step's masks.  With the log on, every
 * forward records one entry per site; the mask of an entry is s2st_dropout_f32(ones, y, n, 1, p, seed, 0) over its element
 * geometry:
 *   S2ST_SITE_LINEAR  y = dropout(act(x W^T + b))            dims = {rows, N}: element (row, n) -> row * N + n, rows in [B][T] order
 *   S2ST_SITE_ATTN    dropout on the attention probabilities  dims = {B, H, T, S, ld}: ((b * H + h) * T + t) * ld + s
 *   S2ST_SITE_ROWS    dropout after the position add          dims = {rows, C}
 *   S2ST_SITE_NORM    dropout after BatchNorm (+ tanh / ReLU) dims = {rows = B * T, C}
 * ctx names the place ("enc.pe", "enc.L3", "dec.prenet", "dec.pe", "dec.L0", "post", "asr.pe", "asr.L0", "st...", "s2t...",
 * "enc.prenet"); ordinal counts the sites of the same kind inside that place in forward order. */
#define S2ST_SITE_LINEAR 1
#define S2ST_SITE_ATTN 2
#define S2ST_SITE_ROWS 3
#define S2ST_SITE_NORM 4
typedef struct s2st_dropout_site {
  uint64_t seed;
  int32_t kind;
  int32_t ordinal;
  float p;
  int32_t reserved;
  int64_t dims[5];
  char ctx[24];
} s2st_dropout_site;
/* allow = 0: the engine never creates its second stream (inference twins that only decode); before the first forward */
int s2st_engine_allow_side_stream(s2st_engine* e, int32_t allow);
int s2st_engine_site_log(s2st_engine* e, int32_t on);
int32_t s2st_engine_site_log_get(const s2st_engine* e, s2st_dropout_site* out, int32_t cap);
/* optional second bf16 arena (param_floats elements): every training forward stores W^T of each 2-D
 * weight there (on the engine's second stream) so the data-gradient GEMMs read K-contiguous operands */
int s2st_engine_bind_bf16_transposed(s2st_engine* e, uint16_t* params_bf16_t);
/* workspace (floats) one forward+backward of this batch geometry needs */
int64_t s2st_engine_workspace_floats(s2st_engine* e, const s2st_batch* geometry);
/* forward (+ losses if tgt != NULL).  Activations live in `workspace` until the next call. */
int s2st_engine_forward(s2st_engine* e, const s2st_batch* b, const s2st_outputs* out,
                        float* workspace, int64_t workspace_floats, void* stream);
/* backward of the last forward: grads += gscale * dLoss/dparams. `upto` = -1 runs the whole
 * tape; otherwise runs tape segments [from previous call, upto) so the caller can interleave
 * gradient all-reduce of finished arena ranges (see s2st_engine_num_segments). */
int s2st_engine_backward(s2st_engine* e, float gscale, int32_t segment, void* stream);
int32_t s2st_engine_num_segments(const s2st_engine* e);
/* The engine's second HIP stream (weight-gradient GEMMs, parameter-gradient reduces) or NULL.  After
 * s2st_engine_backward(segment i) returns, the gradients of segment i are complete once BOTH the
 * caller's stream and this stream have drained what was enqueued so far: a gradient all-reduce on
 * another stream must wait on both (the data-path stream itself joins only after the last segment). */
void* s2st_engine_side_stream(const s2st_engine* e);
/* after segment i has run, gradients in arena range [lo, hi) are final */
int s2st_engine_segment_range(const s2st_engine* e, int32_t i, int64_t* lo, int64_t* hi);
/* The optimizer update of the bound arenas (trainer.py:838-873 + adam.py:163-239, as s2st_adam_f32 with zero_grad = 1)
 * OVERLAPPED with the next forward: the update runs in n_chunks pieces of the arena on the engine's second stream, behind
 * everything enqueued on `stream` so far; the next s2st_engine_forward on the data-path stream waits for a piece right
 * before the first op that reads parameters of it (the arena is in forward-use order), so the HBM-bound update hides behind
 * the forward's first layers.  Anything ELSE that reads the parameters (host copies, a checkpoint, an all-reduce of
 * parameters) first calls s2st_engine_wait_optimizer(e, its stream).  exp_avg / exp_avg_sq: the Adam moments, arena layout;
 * sumsq_parts / n_parts: s2st_sumsq_parts_f32 of the gradient arena; write_bf16: refresh the bound bf16 arena as well.
 * Without a second stream the update simply runs on `stream`. */
int s2st_engine_adam_overlapped(s2st_engine* e, float* exp_avg, float* exp_avg_sq, const float* sumsq_parts, int32_t n_parts, float gmul, const float* gmul_dev, float max_norm, float lr, float beta1, float beta2, float eps, float wd, int32_t step, float* gnorm_out, int32_t* skipped, int32_t write_bf16, int32_t n_chunks, void* stream);
int s2st_engine_wait_optimizer(s2st_engine* e, void* stream);

/* ---- AR inference (config 5): fairseq/speech_generator_for_s2st.py:46-110 drives these.
 * decode_begin runs the encoder on `b` (eval mode; out->enc_out / tap0 / tap1 receive the encoder
 * outputs) and fills the caller-owned `state` (self-attention K/V caches for max_steps frames + the
 * static cross-attention K/V of every decoder layer; with S2ST_DECODE_KV_BF16=1 in the bf16 mode also bf16 copies of
 * those rows, which the steps' cross-attention then reads).  decode_step consumes the previous output
 * frame prev [B][out_dim] (zeros at step 0), appends to the caches and returns feat_out [B][out_dim]
 * (pre-post-net features), eos_prob [B] = sigmoid(stop logit) and, optionally, the head-averaged
 * cross-attention of the last layer attn_out [B][E].  `pos` [B] = positional-table row of this step
 * (step + 2: fairseq's padding_idx + 1 offset).  Prenet dropout is always on (tacotron2.py:95-98):
 * `seed` keys its mask.  postnet_eval = feat + postnet(feat) with BatchNorm running statistics. */
int64_t s2st_engine_decode_state_floats(const s2st_engine* e, int32_t B, int32_t E, int32_t max_steps);
int s2st_engine_decode_begin(s2st_engine* e, const s2st_batch* b, const s2st_outputs* out, float* state,
                             int64_t state_floats, int32_t max_steps, float* workspace,
                             int64_t workspace_floats, void* stream);
/* several batches decoded as ONE merged batch (round 6): row_map [B] (device int32) = for every row of the merged batch the
 * row of its OWN batch -- the always-on Prenet dropout (tacotron2.py:95-98) keys its mask by (seed, row, column), so the merged
 * run draws for every utterance the mask its own batch would have drawn (speech_generator_for_s2st.py:83-103 run batch after
 * batch).  Call after decode_begin (which clears it); NULL: rows count from 0. */
int s2st_engine_decode_row_map(s2st_engine* e, const int32_t* row_map);
int s2st_engine_decode_step(s2st_engine* e, int32_t step, const float* prev, const int32_t* pos,
                            const int32_t* self_klen /* [B] or NULL: self-attention keys per utterance */,
                            uint64_t seed, float* feat_out, float* eos_prob, float* attn_out, float* workspace,
                            int64_t workspace_floats, void* stream);

/* The same step in a form a HIP graph can replay (round 5: config 5 was bound by the host's ~15 k launches per batch).
 * Everything that changes from step to step is read from DEVICE memory, so one captured step serves a whole run:
 *   step      [1]  the step the next replay computes (self-attention: keys 0 .. step, cache row step);
 *   seeds     [8]  the dropout seeds of the step's sites (prenet layers), in launch order;
 *   cur_feat  [B][out_dim]  in: the previous step's features (the zero frame before step 0); out: this step's;
 *   cur_eos   [B], cur_attn [B][E] (or NULL)  this step's stop probabilities / head-averaged alignment;
 *   pe_cur    [dec_dim]  alpha * PE[step + 2], the row every utterance adds at this step.
 * decode_replay_begin sets the state for step 0 (seed0 = the seed decode_step would get at step 0; step s gets seed0 + s).
 * decode_step_replay enqueues one step (S2ST_ERR_SHAPE where the step is not made of the skinny / fused forms: the caller
 * falls back to decode_step).  decode_replay_commit is the stop rule of s2st_decode_stop_update_i32 for the step, copies the
 * step's outputs to row `step` of the run's buffers (feat_all [max_iter][B][out_dim], eos_all [max_iter][B], attn_all
 * [max_iter][B][E] or NULL), prepares pe_cur / seeds for the next step and advances `step`.  Results are those of
 * decode_step + s2st_decode_stop_update_i32 called step by step, bit for bit (tests/test_inference.py). */
typedef struct {
  int32_t* step;
  uint64_t* seeds;
  float* cur_feat;
  float* cur_eos;
  float* cur_attn;
  float* pe_cur;
} s2st_decode_replay;
int32_t s2st_engine_decode_replay_supported(const s2st_engine* e); /* after decode_begin: 1 if the run's steps can take this form */
int s2st_engine_decode_replay_begin(s2st_engine* e, const s2st_decode_replay* r, uint64_t seed0, void* stream);
int s2st_engine_decode_step_replay(s2st_engine* e, const s2st_decode_replay* r, const int32_t* self_klen, float* workspace,
                                   int64_t workspace_floats, void* stream);
int s2st_engine_decode_replay_commit(s2st_engine* e, const s2st_decode_replay* r, uint64_t seed0, float thr, int32_t max_iter,
                                     int32_t* finished, int32_t* out_lens, int32_t* klen_next, int32_t* n_done,
                                     float* feat_all, float* eos_all, float* attn_all, void* stream);
int s2st_engine_postnet_eval(s2st_engine* e, const float* feat, int32_t B, int32_t D, float* post_out,
                             float* workspace, int64_t workspace_floats, void* stream);

/* helpers of the generator and the Griffin-Lim vocoder (fairseq/models/text_to_speech/vocoder.py:84-144) */
int s2st_argmax_dim1_f32(const float* x, int64_t* idx, int32_t B, int32_t E, int32_t D, void* stream);
int s2st_affine_cols_f32(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int32_t C, void* stream);
int s2st_exp_transpose_f32(const float* x, float* y, int32_t T, int32_t C, void* stream);
int s2st_clamp_min_f32(float* x, int64_t n, float lo, void* stream);
int s2st_gl_polar_f32(const float* mag, const float* ang, float* X, int32_t F, int32_t T, void* stream);
int s2st_gl_project_f32(const float* mag, const float* Y, float* X, int32_t F, int32_t T, void* stream);
int s2st_reflect_pad_f32(const float* x, float* y, int32_t n, int32_t pad, void* stream);
int s2st_gl_overlap_add_f32(const float* frames, const float* wsq, float* wave, int32_t T, int32_t n_fft, int32_t hop, int32_t n_out, void* stream);

/* Batched Griffin-Lim on the bf16 matrix cores (vocoder.py:100-123 for every utterance of a batch at once):
 * U utterances of tl[u] <= Tmax frames, rows r = u * Tmax + t.  fp32 rows are stored for the GEMM as bf16
 * [hi | lo | hi] (3 K) against constant bases [hi | hi | lo], i.e. the bf16x3 split folded into the
 * contraction.  Spectra: [rows][re(F) pad | im(F) pad], halves Fp apart (Fp % 4 == 0).
 *   polar_split: Xs = split(mag * exp(i a)), a = aux[rows][F] (from_spectrum 0) or angle(aux[rows][2 Fp]) (1)
 *   frame_split: As[rows][3][n_fft] = split(frames of the reflect-padded waves [U][Lw]) */
int s2st_gl_polar_split_f32(const float* mag, const float* aux, int32_t from_spectrum, const int32_t* tl, void* Xs, int32_t U, int32_t F, int32_t Fp, int32_t Tmax, void* stream);
int s2st_gl_frame_split_f32(const float* wave, const int32_t* tl, void* As, int32_t U, int32_t Tmax, int32_t hop, int32_t n_fft, int32_t Lw, void* stream);
/* Griffin-Lim with FFTs (round 4; replaces, for n_fft a power of two in 256 ... 2048, the dense Fourier-basis contractions
 * above).  The reference's analysis basis [Re; Im] fft(eye(n_fft)) * window is rfft(window * frame)
 * (fairseq/data/audio/audio_utils.py:226-231, 259-271) and its synthesis basis pinverse(n_fft / hop * basis)^T * window is
 * window * (hop / n_fft) * irfft (fairseq/models/text_to_speech/vocoder.py:59-62, 82-86).  X: complex spectra
 * [U * Tmax][n_fft / 2 + 1] (re, im interleaved); tl [U] frames per utterance; win [n_fft]; tw [n_fft] complex exp(-2 pi i j / n_fft).
 * s2st_gl_polar_c_f32: X = mag * exp(i ang) (vocoder.py:101-103).  s2st_gl_stft_project_f32: reflect-pad + frame + window +
 * rfft of wave [U][Lw], then X = mag * Y / |Y| (vocoder.py:104-107).  s2st_gl_istft_frames_f32: synthesis frames
 * [U * Tmax][n_fft] for s2st_gl_overlap_add_b_f32. */
/* First block of the HuBERT / wav2vec 2.0 feature extractor (fairseq/models/wav2vec/wav2vec2.py:777-783, 806-814:
 * Conv1d(1, C, k, stride, bias=False) -> Fp32GroupNorm(C, C) -> GELU) on wave [B][N]: y / y_bf16 [B][T][C] (channel-last;
 * either may be NULL), T = (N - k) / stride + 1, statistics over the T frames of each (utterance, channel).  stats: scratch
 * of s2st_hubert_conv0_stats_floats_i64(B, T, C) floats.  k <= 16, stride <= 8, C % 4 == 0. */
int64_t s2st_hubert_conv0_stats_floats_i64(int32_t B, int32_t T, int32_t C);
int s2st_hubert_conv0_gn_gelu_f32(const float* wave, const float* w, const float* gamma, const float* beta, float* y, uint16_t* y_bf16, float* stats, int32_t B, int32_t N, int32_t T, int32_t C, int32_t k, int32_t stride, float eps, void* stream);
/* One decoding step's attention (fairseq/modules/multihead_attention.py:194-385 with incremental_state: one query per
 * utterance against the cached keys / values, key padding mask = klen, softmax, weighted values; head-averaged weights of the
 * alignment layer into attn_mean [B][S], zeroed here).  q [B][ldq] (head h at columns h * dh); caches: row s of utterance b at
 * cache + b * kbs + s * ldk (elements), fp32 or -- kv_bf16 = 1, head widths 64 / 128, static rows only -- bf16.  k_new / v_new
 * [B][ld_new] (optional, fp32 caches): this step's rows, stored as row pos_new of the caches before attending (saved-state
 * update, :300-318). */
int s2st_decode_attn_f32(const float* q, int64_t ldq, void* k_cache, void* v_cache, int64_t ldk, int64_t kbs, const int32_t* klen, int32_t nkeys, int32_t B, int32_t H, int32_t dh, float scale, float* o, int64_t ldo, float* attn_mean, int32_t S, const float* k_new, const float* v_new, int64_t ld_new, int32_t pos_new, int32_t kv_bf16, void* stream);
/* The AR generator's stop rule on the device (fairseq/speech_generator_for_s2st.py:84-99): after decoding step `step`,
 * finished |= eos_prob > thr, out_lens of the utterances that just finished = step + 1 (initial value max_iter = "still
 * running"), klen_next = the key lengths the NEXT step's self-attention masks with (:84-85), n_done[step] = how many are
 * finished.  The host reads n_done a few steps late and drops the steps it ran past the stop: no synchronisation per step. */
int s2st_decode_stop_update_i32(const float* eos_prob, float thr, int32_t step, int32_t max_iter, int32_t B, int32_t* finished, int32_t* out_lens, int32_t* klen_next, int32_t* n_done, void* stream);
int s2st_gl_fft_supported_i32(int32_t n_fft);
int s2st_gl_polar_c_f32(const float* mag, const float* ang, const int32_t* tl, float* X, int32_t U, int32_t F, int32_t Tmax, void* stream);
/* initial phases (vocoder.py:101-102) from the uniform draws themselves: uniform = the doubles numpy's generator produced, utterance
 * u's [F][T_u] block at uniform + offsets[u]; X = mag * exp(i wrap(2 pi u)).  uniform == NULL: the draws come from the device's
 * counter-based generator (seed): same distribution, not numpy's stream. */
/* numpy's legacy generator (MT19937, np.random.random_sample / rand: one double per two 32-bit outputs) -- what the
 * reference's GriffinLim draws on the host (vocoder.py:101-102), draw for draw --
 * on the HOST, several threads (csrc/mt19937_host.cpp): `state` = 624 key words + position (625 uint32),
 * out = n doubles in host memory (NULL: states only), bounds_out (optional) = (threads + 1) records of 625 words: the state
 * in front of double n * t / threads, t = 0 .. threads.  Thread t reaches its share by running the recurrence alone. No GPU
 * work: replaces the host-side np.random.rand of vocoder.py:101-102 (0.37 G draws/s on one core) draw for draw. */
int s2st_mt19937_host_doubles(const uint32_t* state, int64_t n, double* out, uint32_t* bounds_out, int32_t threads);
/* x <- exp(x) in place (vocoder.py:139 for a padded batch of log-mel frames) */
int s2st_exp_inplace_f32(float* x, int64_t n, void* stream);
int s2st_gl_polar_u_f32(const float* mag, const double* uniform, const int64_t* offsets, const int32_t* tl, uint64_t seed, float* X, int32_t U, int32_t F, int32_t Tmax, void* stream);
int s2st_gl_stft_project_f32(const float* wave, const int32_t* tl, const float* win, const float* tw, const float* mag, float* X, int32_t U, int32_t Tmax, int32_t n_fft, int32_t hop, int32_t Lw, void* stream);
int s2st_gl_istft_frames_f32(const float* X, const int32_t* tl, const float* win, const float* tw, float* frames, int32_t U, int32_t Tmax, int32_t n_fft, int32_t hop, void* stream);
int s2st_gl_overlap_add_b_f32(const float* frames, const float* wsq_all, const int64_t* wsq_off, const int32_t* tl, float* wave, int32_t U, int32_t Tmax, int32_t n_fft, int32_t hop, int32_t Lw, void* stream);
/* s2st_gl_istft_frames_f32 + s2st_gl_overlap_add_b_f32 as ONE launch (the synthesis frames stay in LDS): X [U * Tmax][F]
 * complex -> wave [U][Lw], vocoder.py:84-98 (conv_transpose1d with the pseudo-inverse basis, window-sum-square division
 * above `tiny`, * n_fft / hop, n_fft / 2 trimmed at both ends); equal to the two calls to fp32 rounding. */
int s2st_gl_istft_ola_f32(const float* X, const int32_t* tl, const float* win, const float* tw, const float* wsq_all, const int64_t* wsq_off, float* wave, int32_t U, int32_t Tmax, int32_t n_fft, int32_t hop, int32_t Lw, void* stream);

/* AR decoding (speech_generator_for_s2st.py:76-110: one decoder step for the B utterances of a batch): skinny
 * y[M][N] = f(x[M][K] W[N][K]^T + bias) (+ resid), M <= 16, K % 32 == 0; x fp32 (rounded to bf16 in registers like the
 * operand copies of s2st_gemm_f32), W bf16; act 0 / 1 relu / 2 gelu, dropout mask from (seed, m * N + n) */
int s2st_gemm_skinny_f32(const float* x, int64_t ldx, const void* w_bf16, int64_t ldw, float* y, int64_t ldy, const float* bias, int32_t act, float drop_p, uint64_t seed, const float* resid, int64_t ldr, int32_t M, int32_t N, int32_t K, void* stream);

/* the same with the decoder's pre-LayerNorm fused in front: y = f(LayerNorm(x; gamma, beta, eps) W^T + bias), K % 64 == 0 */
int s2st_ln_gemm_skinny_f32(const float* x, int64_t ldx, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w_bf16, int64_t ldw, float* y, int64_t ldy, const float* bias, int32_t act, int32_t M, int32_t N, int32_t K, void* stream);

/* Host-side: max-tokens batching of length-sorted indices (fairseq/data/data_utils_fast.pyx:20-100,
 * batch_by_size_vec; the reference builds it as a Cython extension).  num_tokens in index order, batch_ends: n + 1
 * int32 slots; returns the number of batches (batch k = positions [batch_ends[k-1], batch_ends[k])), or a
 * negative S2ST_ERR_* (an item longer than max_tokens: S2ST_ERR_SHAPE). */
int64_t s2st_batch_by_size(const int64_t* num_tokens, int64_t n, int64_t max_tokens, int64_t max_sentences, int32_t bsz_mult, int32_t* batch_ends);

/* MCD evaluation (examples/s2s_trans/tasks/s2s_translation.py:414-552): batched DTW over the padded
 * [B][M][N] distance tensor (shapes [B][2] = (m, n) per element, or NULL), RMS feature distance, MFCC glue */
int s2st_dtw_f32(const float* dist, const int32_t* shapes, int32_t B, int32_t M, int32_t N, float* cumdist, int32_t* backptr, int32_t* pathmap, void* stream);
int s2st_rms_dist_f32(const float* x1, const float* x2, float* out, int32_t m, int32_t n, int32_t D, int64_t ldo, void* stream);
int s2st_power_spec_f32(const float* Y, float* P, int32_t T, int32_t F, void* stream);
int s2st_log_offset_f32(float* x, int64_t n, float eps, void* stream);

/* Aux ASR (which = 0) / ST (which = 1) text decoder, forward only, over an encoder tap: what the reference runs
 * when fairseq_cli/generate_for_s2st.py:107-111 swaps model.decoder for model.aux_{asr,st}_decoder and
 * SequenceGenerator calls decoder.forward(tokens, encoder_out) (examples/s2s_trans/models/s2st_transformer.py:483-578,
 * fairseq/models/transformer/transformer_decoder.py:253-378).  tap [Bb][E][enc_dim] = the head's normalised tap of the
 * encoder (outputs tap0 / tap1), one copy per hypothesis; prev_tokens [Bb][L] int64; positions [Bb][L] int32
 * (make_positions); lens [Bb] = valid tokens per row; pe = sinusoidal table of width asr_dim / st_dim;
 * logits_out [Bb][L][V].  s2st_engine_aux_decode_workspace gives the workspace floats for (Bb, L, E). */
int64_t s2st_engine_aux_decode_workspace(s2st_engine* e, int32_t which, int32_t Bb, int32_t L, int32_t E);
int s2st_engine_aux_decode(s2st_engine* e, int32_t which, const float* tap, const int32_t* enc_lens,
                           const int64_t* prev_tokens, const int32_t* positions, const int32_t* lens, const float* pe,
                           int32_t Bb, int32_t L, int32_t E, float* logits_out, float* workspace,
                           int64_t workspace_floats, void* stream);

/* The same decoder STEP BY STEP with key / value caches -- what fairseq's SequenceGenerator does through incremental_state
 * (fairseq/sequence_generator.py:189-571: decoder.forward(tokens, incremental_state) on the last token only, and
 * reorder_incremental_state(new_order) after every step's beam selection; fairseq/modules/multihead_attention.py:261-299, 387-403;
 * fairseq/models/transformer/transformer_decoder.py:281-326 takes the LAST position when an incremental state is given).
 * state: caller-owned, s2st_engine_aux_inc_state_floats(which, Bb, E, max_len) floats -- two copies of the self-attention caches
 * (a reorder gathers from one into the other) and the layers' static encoder K | V projections.  begin: tap [Bb][E][enc_dim] (one
 * copy per hypothesis), enc_lens [Bb].  step s: tokens [Bb] = the hypotheses' last tokens (int64), reorder [Bb] (int32, or NULL)
 * = hypothesis b continues old hypothesis reorder[b], positions [Bb] = s + 2 (prefixes hold no padding), pe as above;
 * logits_out [Bb][V] = the last position's logits.  A hypothesis costs O(L) per step (the prefix form above: O(L^2)). */
int64_t s2st_engine_aux_inc_state_floats(const s2st_engine* e, int32_t which, int32_t Bb, int32_t E, int32_t max_len);
int64_t s2st_engine_aux_inc_workspace(const s2st_engine* e, int32_t which, int32_t Bb, int32_t E);
int s2st_engine_aux_inc_begin(s2st_engine* e, int32_t which, const float* tap, const int32_t* enc_lens, int32_t Bb, int32_t E,
                              int32_t max_len, float* state, float* workspace, int64_t workspace_floats, void* stream);
int s2st_engine_aux_inc_step(s2st_engine* e, int32_t which, int32_t step, const int64_t* tokens, const int32_t* reorder,
                             const int32_t* positions, const float* pe, float* logits_out, float* workspace,
                             int64_t workspace_floats, void* stream);

/* ---- frozen HuBERT front end of config 4 (--use-hubert): fairseq/models/hubert/hubert.py:412-461,
 * 518-534 (extract_features, eval, mask=False) with wav2vec2.py:736-905.  The handle is an
 * s2st_engine in "hubert mode": parameters are enumerated / bound with s2st_engine_param_info,
 * s2st_engine_bind(params, NULL, NULL) and s2st_engine_bind_bf16; conv weights are stored in GEMM
 * layout [O][k][I] and the weight-normed pos_conv as its effective weight [G][E/G][k][E/G] (the host
 * wrapper converts from the reference state_dict).  Forward only. */
typedef struct {
  int32_t n_conv;
  int32_t conv_dim[8], conv_k[8], conv_stride[8];
  int32_t embed, layers, heads, ffn, conv_pos, conv_pos_groups;
  int32_t precise;
} s2st_hubert_config;
int s2st_hubert_create(const s2st_hubert_config* cfg, s2st_engine** out);
int32_t s2st_hubert_out_frames(const s2st_engine* e, int32_t n_samples);
int64_t s2st_hubert_workspace_floats(s2st_engine* e, int32_t B, int32_t N);
/* wave [B][N] fp32 (zero right-padded), frame_lens [B] = valid output frames per utterance
 * (hubert.py:400-410 applied to the sample padding mask) -> out [B][T'][embed] fp32 */
int s2st_hubert_forward(s2st_engine* e, const float* wave, const int32_t* frame_lens, int32_t B, int32_t N,
                        float* out, float* workspace, int64_t workspace_floats, void* stream);

/* ======================================================================================
 * Gradient exchange: SUM all-reduce over an RCCL communicator, one process per GPU.  Replaces the bucketed NCCL
 * all-reduce torch DDP runs for the reference (fairseq/models/distributed_fairseq_model.py:58-67); the trainer hands
 * over contiguous ranges of the flat gradient arena as the backward finishes them (s2st_engine_segment_range) and
 * scales by 1 / sum(sample_size) in s2st_adam_f32.  RCCL is bound at run time (the copy the process already carries,
 * else $S2ST_RCCL_LIB, else librccl.so.1): libs2st_hip.so has no link-time dependency on it.
 *   rank 0: s2st_comm_unique_id(id) -> id (128 bytes) to every rank out of band -> all ranks: s2st_comm_init
 * ====================================================================================== */
typedef struct s2st_comm s2st_comm;
int s2st_comm_available(void);
int s2st_comm_unique_id(void* id128);
int s2st_comm_init(const void* id128, int32_t world, int32_t rank, s2st_comm** out);
/* buf[0..n) <- sum over ranks (in place), ordered on `stream` */
int s2st_allreduce_sum_f32(s2st_comm* comm, float* buf, int64_t n, void* stream);
int s2st_comm_destroy(s2st_comm* comm);

/* Measurement aid (bench.py roofline leg).  While enabled, the dominant kernels (GEMMs, split-K combine, fused
 * attention, LayerNorm, optimizer) are launched with a start / stop event attached to the dispatch itself, on the
 * stream they run on: the elapsed time of such a pair is the dispatch's own begin -> end (what rocprofv3
 * --kernel-trace reports).  s2st_profile_report writes one line per kernel instantiation,
 * "tag\tlaunches\ttotal_us\twork\twork2\n" (work = as-launched FLOPs of a GEMM / attention launch, or the bytes an
 * HBM-bound kernel has to move), clears the registry and returns the text length (-1: buffer too small). */
int s2st_profile_enable(int32_t enable);
/* 16 hex digits identifying the kernel / engine sources this library was built from (sha256 prefix over csrc/ and
 * include/, __graft_entry__.source_hash): measurements stored under profiles/ carry it, bench.py reports a stored PMC
 * traffic figure only when it matches */
int s2st_source_hash(char* out, int32_t cap);
int64_t s2st_profile_report(char* out, int64_t cap);
/* Same registry as one line per dispatch in launch order: "tag\tstream\tstart_us\tdur_us\n" (stream = index in order of
 * first use, start on the GPU clock relative to the first dispatch).  Clears the registry. */
int64_t s2st_profile_timeline(char* out, int64_t cap);

int s2st_version(void);
/* 1 when the library was built with -DS2ST_EXPERIMENTAL: the measured-and-not-chosen GEMM forms (persistent tile walk,
 * stream-K, 256 x 128 tiles: S2ST_GEMM_PERSIST=2/3, S2ST_GROUP_ONESHOT=0, S2ST_GROUP_TILE=256, S2ST_GEMM_STREAMK) and the
 * timing-only switch S2ST_TIMING_SKIP_WGRAD exist only there; the product build (0) ignores those switches */
int s2st_experimental_build(void);
/* number of HIP devices visible (0 = none: every compute entry point then fails) */
int s2st_device_count(void);
#ifdef __cplusplus
}
#endif
#endif /* S2ST_HIP_H */
