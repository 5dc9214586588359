// Host-side launcher interface of the s2st HIP kernels (internal C++ API; the C ABI in
// include/s2st_hip.h wraps these).  All launchers are stateless, stream-ordered, take
// caller-owned device buffers and return 0 or a negative error code (no exceptions).
#pragma once
#include <cstdlib>
// The library's ONE reader of the process environment: every switch of DESIGN.md section 4's table goes through these three
// (so that `grep getenv csrc/` finds this place and nothing else, and a switch cannot be read in two spellings).
static inline const char* s2st_env_str(const char* name) { return getenv(name); }
static inline int s2st_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline bool s2st_env_on(const char* name) { return s2st_env_int(name, 0) != 0; }

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
  o.p = p; o.kmajor = 1; o.dtype = S2ST_F32; o.sp.ld = ld; o.sp.bs = 0; o.sp.per = 0; o.sp._pad = 0;
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
  o.p = p; o.sp.ld = ld; o.sp.bs = 0; o.sp.per = 0; o.sp._pad = 0; o.zo = o.zi = 0; o.h = nullptr;
  return o;
}
typedef uint16_t bf16raw;
inline GemmOperand gemm_rowmajor(const bf16raw* p, long ld) {
  GemmOperand o = gemm_rowmajor((const float*)nullptr, ld);
  o.p = p; o.dtype = S2ST_BF16;
  return o;
}
inline GemmOperand gemm_colmajor(const bf16raw* p, long ld) {
  GemmOperand o = gemm_rowmajor(p, ld);
  o.kmajor = 0;
  return o;
}
inline GemmEpilogue gemm_epi_default() {
  GemmEpilogue e;
  e.alpha = 1.f; e.bias = nullptr; e.act = 0; e.drop_p = 0.f; e.seed = 0; e.resid = nullptr;
  e.accumulate = 0;
  e.mask_y = nullptr; e.mask_scale = 1.f; e.colsum = nullptr; e.colsum_part = nullptr; e.bias_zo = 0;
  return e;
}

// up to S2ST_GROUP_MAX bf16 problems with the same operand layouts in ONE persistent launch (gemm_bf16.hip): batch 1,
// no split-K; total = tile0[n] tiles of the launcher's tile size
#define S2ST_GROUP_MAX 8
struct GemmGroup {
  int n, total;
  int tile0[S2ST_GROUP_MAX + 1];
  GemmArgs g[S2ST_GROUP_MAX];
  // stream-K form (sk != 0): the K-steps of ALL tiles are dealt out evenly to the workgroups; a workgroup whose range
  // ends inside a tile leaves its partial accumulators in sk_part[w] and raises sk_flag[w] = epoch, the workgroup
  // that reaches the tile's last K-step adds them and runs the epilogue.  sk_ctr: 8 per-XCD ticket counters + a
  // completion counter (zero before the first launch; the last workgroup of a launch resets them).
  int sk, epoch;
  int xcd_global;  // one-shot grouped form: each XCD owns one contiguous run of the CONCATENATED tile list
  int* sk_ctr;
  int* sk_flag;
  float* sk_part;
};
// scratch for stream-K launches on `st` (nullptr: none -- the launcher then keeps whole tiles per workgroup);
// floats >= S2ST_STREAMK_SCRATCH_FLOATS; the first 16 ints must be zero at bind time
#define S2ST_STREAMK_MAX_WGS 512
#define S2ST_STREAMK_SCRATCH_FLOATS (1024 + (long)S2ST_STREAMK_MAX_WGS * 128 * 128)
void s2st_gemm_streamk_bind(hipStream_t st, float* scratch, long floats);
void s2st_gemm_streamk_unbind_all();
// true if g can join a group (aligned bf16 operands, plain strides, batch 1, >= 128 x 128 of output)
bool s2st_gemm_group_ok(const GemmArgs& g);
int s2st_gemm_bf16_group(const GemmArgs* list, int n, hipStream_t st);
int s2st_gemm(GemmArgs g, hipStream_t st, int* tile_out = nullptr /* bf16 path: tile rows * 1000 + tile columns */);
int s2st_gemm_bf16(GemmArgs g, hipStream_t st, int* tile_out);
// skinny-M (<= 16 rows) y = f(x W^T + b) (+ resid) with fp32 x converted in registers (AR decoding)
int s2st_gemm_skinny(const float* A, long lda, const bf16raw* W, long ldw, float* C, long ldc, const float* bias, int act,
                     float drop_p, uint64_t seed, const float* resid, long ldr, int M, int N, int K, hipStream_t st,
                     const float* ln_g = nullptr, const float* ln_b = nullptr, float ln_eps = 1e-5f /* optional fused LayerNorm of x */,
                     const uint64_t* seed_ptr = nullptr /* replayable decode step: the seed is read from device memory */,
                     const int* row_map = nullptr /* [M]: the row whose dropout mask row m draws (merged decode batches) */);
// rows a skinny launch takes (row blocks of 16 on grid.y): up to four batches of 64 utterances decoded as one merged batch
#define S2ST_SKINNY_MAX_ROWS 256
// gemm_bf16_w4.hip: the 4-wave early-release ring form (2 - 3 workgroups per CU) for the short-K products of a step;
// g / grp as prepared by s2st_gemm_bf16 / s2st_gemm_bf16_group for the tile (bm, bn)
int s2st_gemm_bf16_w4(const GemmArgs& g, int bm, int bn, dim3 grid, hipStream_t st);
int s2st_gemm_bf16_w4_group(const GemmGroup& grp, hipStream_t st);
int s2st_gemm_bf16_w4_preload(hipStream_t st);
// 256 x 256 four-phase form (gemm_bf16_p4.hip): both operands K-contiguous, chosen by p4_pick() in gemm_bf16.hip
int s2st_gemm_bf16_p4(const GemmArgs& g, dim3 grid, hipStream_t st);
int s2st_gemm_bf16_p4_preload(hipStream_t st);
int s2st_gemm_bf16_preload(hipStream_t st);  // load every instantiation (empty launches)  // gemm_bf16.hip (both operands bf16)
void s2st_profile_enable_impl(int on);                 // per-dispatch timing registry (s2st_prof.h, gemm.hip)
long s2st_profile_report_impl(char* out, long cap, int mode = 0);

// ---------------------------------------------------------------------------------------
// row ops (rowops.hip)
// ---------------------------------------------------------------------------------------
int s2st_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int rows, int cols, float eps, hipStream_t st,
                       uint16_t* yh = nullptr /* optional bf16 copy of y */);
// dx (=|+=) ...; dgamma += , dbeta += (per-block partials in `scratch`, then a reduce kernel)
int s2st_layernorm_bwd_blocks(int rows, int cols);  // scratch floats = blocks * (dph ? 3 : 2) * cols
// column-sum partials of several layer-norm backward passes (phase 3 below), folded by ONE launch in a fixed order
#define S2ST_LNFOLD_MAX 24
// (part2 / nblocks2: a second array of partial rows for the SAME outputs, summed behind the first -- the second
//  utterance-half chain's partials, engine.cpp S2ST_CHAINS=2; s2st_fold_add attaches it by itself when the outputs of a
//  new entry equal those of one already in the table, because two entries with one output would race in the fold)
struct s2st_lnfold_item { const float* part; float *dgamma, *dbeta, *dbias; int nblocks, cols, nout; const float* part2; int nblocks2; };
struct s2st_lnfold_table { int n; int blk0[S2ST_LNFOLD_MAX + 1]; s2st_lnfold_item item[S2ST_LNFOLD_MAX]; };
int s2st_lnfold_add(s2st_lnfold_table& t, const float* part, int rows, int cols, int nout, float* dgamma, float* dbeta,
                    float* dbias);
int s2st_layernorm_bwd_fold(const s2st_lnfold_table& t, hipStream_t st);
// any other [nblocks][nout][cols] array of partial rows: out_i[c] += sum_b part[b][i][c], b in order (out_i may be null)
int s2st_fold_add(s2st_lnfold_table& t, const float* part, int nblocks, int cols, int nout, float* out0, float* out1,
                  float* out2);
// dph: optional fused backward prologue of the linear+dropout layer that produced x (rowops.hip)
int s2st_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                       const float* rstd, float* dx, int dx_accumulate, float* dgamma,
                       float* dbeta, float* scratch, int rows, int cols, hipStream_t st, int phase = 0,
                       uint16_t* dph = nullptr, float drop_p = 0.f, uint64_t seed = 0, float* dbias = nullptr);

// attention probabilities: p = softmax(s + masks) rows of [B, H, T, S(ld)]
//   key mask: col >= klen[b] -> -inf ; causal: col > row -> -inf
//   pd (optional, drop_p > 0) = dropout(p)
int s2st_softmax_fwd(const float* s, float* p, float* pd, const int* klen, int B, int H, int T,
                     int S, int ld, int causal, float drop_p, uint64_t seed, hipStream_t st,
                     uint16_t* pdh = nullptr /* optional bf16 copy of dropout(p), pad cols zeroed */);
// ds = p * (dp' - sum(dp' * p)),  dp' = dropmask * dpd ; in place allowed (ds == dpd)
int s2st_softmax_bwd(const float* p, const float* dpd, float* ds, int B, int H, int T, int S,
                     int ld, float drop_p, uint64_t seed, hipStream_t st,
                     uint16_t* dsh = nullptr /* optional bf16 copy of ds, pad cols zeroed */);

// column sums: out[c] (+)= sum_r x[r][c]   (bias gradients)
int s2st_colsum(const float* x, long ld, int rows, int cols, float* out, int accumulate,
                hipStream_t st, float* part = nullptr /* fixed-order sums through this scratch (s2st_colsum_scratch_floats) */,
                int* slabs_out = nullptr /* with part: leave the fold to the caller (s2st_fold_add), return the partial rows */);
long s2st_colsum_scratch_floats(int rows, int cols);
int s2st_colsum_fold(const float* part, int slabs, int cols, float* out, hipStream_t st);  // out[c] += sum_s part[s][c], in order
// out[c] += sum_r x[r][c] for a bf16 matrix (bias gradients of projections whose output gradient only exists in bf16)
int s2st_colsum_bf16(const uint16_t* x, long ld, int rows, int cols, float* out, hipStream_t st);
// the same without atomics (slab partials in `scratch`, s2st_colsum_bf16_scratch_floats floats, folded in a fixed order)
long s2st_colsum_bf16_scratch_floats(int rows, int cols);
int s2st_colsum_bf16_ordered(const uint16_t* x, long ld, int rows, int cols, float* out, float* scratch, hipStream_t st);
// mean over heads of attention probabilities: out[b][s][t] = mean_h p[b][h][t][s]
int s2st_attn_headmean(const float* p, float* out, int B, int H, int T, int S, int ld,
                       hipStream_t st);

// ---------------------------------------------------------------------------------------
// elementwise (elementwise.hip)
// ---------------------------------------------------------------------------------------
// fp32 [rows][cols] (stride ldx) -> bf16 [rows][ldy] (ldy % 4 == 0, pad columns zeroed): the
// bf16 copies the fast GEMM path reads
int s2st_cast_bf16_rows(const float* x, long ldx, uint16_t* y, long ldy, long rows, int cols, hipStream_t st);
// bf16 gradient exchange (optim.hip): round a range of the gradient arena to bf16 / widen the reduced values back
int s2st_grad_pack_bf16(const float* g, uint16_t* out, long n, hipStream_t st);
int s2st_exchange_proxy(const float* bucket, float* scratch, long n, long move_bytes, int wgs, float gbps, hipStream_t st);
int s2st_grad_unpack_bf16(const uint16_t* in, float* g, long n, hipStream_t st);
int s2st_transpose_bf16(const uint16_t* x, uint16_t* y, int R, int C, hipStream_t st);  // [R][C] -> [C][R]
// every registered matrix [rows][cols] at element offset off of x_base -> its transpose at the same offset of y_base,
// one launch (rows, cols multiples of 8; tile0 = running count of 64 x 64 tiles, tile0[n] = grid size)
#define S2ST_TRANSPOSE_MAX 200
struct s2st_transpose_table {
  int n;
  unsigned off[S2ST_TRANSPOSE_MAX];
  unsigned short rows8[S2ST_TRANSPOSE_MAX], cols8[S2ST_TRANSPOSE_MAX];
  unsigned tile0[S2ST_TRANSPOSE_MAX + 1];
};
int s2st_transpose_bf16_batched(const uint16_t* x_base, uint16_t* y_base, const s2st_transpose_table& t, hipStream_t st);
// copy [rows][C] between split-addressed buffers (halo padding, zero-stuffing); C % 4 == 0
int s2st_copy_rows_bf16(const uint16_t* x, Split xsp, uint16_t* y, Split ysp, int rows, int C, hipStream_t st);
// bf16 twin of the fp32 halo image x [B][T + 2 pad][C]: interior converted, halos zero (x's halos are not read)
// plain != 0: x is the plain rows [B * T][C]
int s2st_cast_bf16_halo(const float* x, uint16_t* y, int B, int T, int pad, int C, hipStream_t st, int plain = 0);
// y [B][Th][O] = rows of x [B * Tout][ldx] at u = pad + stride * t, zeros elsewhere (the whole image in one pass)
int s2st_halo_image_bf16(const uint16_t* x, long ldx, uint16_t* y, int B, int Tout, int Th, int O, int pad, int stride,
                         hipStream_t st);
int s2st_copy_rows(const float* x, Split xsp, float* y, Split ysp, int rows, int C,
                   hipStream_t st);
// y[r][c] = a[r][c] * sigmoid(a[r][c + C])   a: [rows][2C] plain ; y rows via split
int s2st_glu_fwd(const float* a, float* y, Split ysp, int rows, int C, hipStream_t st);
int s2st_glu_bwd(const float* a, const float* dy, Split dysp, float* da, Split dasp, int rows,
                 int C, hipStream_t st, uint16_t* dah = nullptr /* optional bf16 twin [rows][ldh] */, long ldh = 0);
// y[r][:] = dropout(scale * x[r][:] + alpha * table[pos[r]][:])   (alpha = *alpha_ptr or 1)
// spk_table / spk_ids (optional): + spk_table[spk_ids[row / T]] before the dropout (speaker conditioning of the encoder)
int s2st_add_pe(const float* x, float* y, const int* pos, const float* table, int rows, int C,
                float scale, const float* alpha_ptr, float drop_p, uint64_t seed, hipStream_t st,
                const float* spk_table = nullptr, const long* spk_ids = nullptr, int T = 0);
// dtable[s] += sum over utterances b with ids[b] == s of the (dropout-masked) first T_sum rows of dy's block b ([B][T][C]);
// fixed summation order (no atomics)
int s2st_speaker_bwd(const float* dy, const long* ids, int B, int T, int T_sum, int C, int n_spk, float drop_p,
                     uint64_t seed, float* dtable, hipStream_t st);
// t2s text encoder (t2s_transformer.py:107-111): y[(b,t)][coff .. coff + Sd) = table[ids[b]] for y rows of stride ld;
// its gradient dtable[s] += sum_{b: ids[b] == s} sum_t dy[(b,t)][coff ..) in index order (no atomics); and
// dx[r][0..C) (+)= g[r][0..C) (the other block of the concatenated gradient)
int s2st_speaker_fill_cols(const float* table, const long* ids, float* y, int B, int T, int ld, int coff, int Sd, hipStream_t st);
int s2st_speaker_cols_bwd(const float* dy, const long* ids, int B, int T, int ld, int coff, int Sd, int n_spk, float* dtable,
                          hipStream_t st);
int s2st_split_cols(const float* g, int ldg, float* dx, int ldx, int rows, int C, int acc, hipStream_t st);
// y[b][0][:] = table[ids[b]] for y [B][T][C]
int s2st_speaker_set_rows(const float* table, const long* ids, float* y, int B, int T, int C, hipStream_t st);
// dalpha += sum dropmask * dy * table[pos]
int s2st_pe_alpha_bwd(const float* dy, const int* pos, const float* table, int rows, int C,
                      float drop_p, uint64_t seed, float* dalpha, hipStream_t st,
                      float* part = nullptr /* per-block partial sums instead of an atomic add (<= 1024 floats) */,
                      int* nparts_out = nullptr);
int s2st_embed_fwd(const long* tokens, const float* table, float* y, int rows, int C, float scale,
                   hipStream_t st);
int s2st_embed_bwd(const long* tokens, const float* dy, float* dtable, int rows, int C, float scale,
                   long pad, hipStream_t st, int V = 0 /* > 0: ordered form -- one thread per (table row, column) adds the
                   matching rows of dy in index order: no atomics */);
// y (+)= a * x * dropmask(p, seed)   (p == 0: plain scaled copy / accumulate)
int s2st_dropout(const float* x, float* y, long n, float a, float p, uint64_t seed, int accumulate,
                 hipStream_t st);
// dz = dy * (y != 0 ? 1/(1-p) : 0)   (backward of dropout(relu(z)) given its OUTPUT y)
int s2st_relu_drop_bwd(const float* dy, const float* y, float* dz, long n, float p,
                       hipStream_t st);
// fused backward prologue of a linear layer (fast mode): dph = bf16(f(dy)) with row stride ldp
// (pad columns zeroed), optional fp32 copy dpre, dbias += colsum(f(dy)).  mode 0: f = id ;
// 1: ReLU+dropout backward from the layer OUTPUT y ; 2: dropout(seed) backward.  N % 4 == 0.
int s2st_linear_bwd_prep(const float* dy, const float* y, const uint16_t* yb /* bf16 y when y == null */, int mode,
                         float p, uint64_t seed, uint16_t* dph, long ldp, float* dpre, float* dbias, int M, int N,
                         hipStream_t st, float* part = nullptr /* bias sums in a fixed order: s2st_linear_bwd_prep_scratch_floats */,
                         int* slabs_out = nullptr /* with part: leave the fold to the caller, return the partial rows */);
long s2st_linear_bwd_prep_scratch_floats(int M, int N, long ldp);
int s2st_axpy(const float* x, float* y, long n, float a, hipStream_t st);  // y += a * x
int s2st_scale(float* x, long n, float a, hipStream_t st);
// conv weight W[O][I][Kw] -> Wf[O][Kw][I] (forward GEMM layout) and, if wd != null,
// Wd[I][Kw-1-j][O] (flipped; data-gradient GEMM layout)
int s2st_conv_w_permute(const float* w, float* wf, float* wd, int O, int I, int Kw, hipStream_t st,
                        uint16_t* wfh = nullptr, uint16_t* wdh = nullptr /* optional bf16 twins */);
// dW[O][I][Kw] += dWf[O][Kw][I]
int s2st_conv_w_unpermute_acc(const float* dwf, float* dw, int O, int I, int Kw, hipStream_t st, int slabs = 1);
// BatchNorm1d in training mode over [rows][C] (rows = ALL B*T positions, padded included,
// tacotron2.py:122-126): two-pass statistics + running-stat update; tmp = 2*C floats
int s2st_bn_stats(const float* x, int rows, int C, float* mean, float* var, float* run_mean,
                  float* run_var, float momentum, float* tmp, hipStream_t st);
// y = dropout([tanh](gamma * xhat + beta)) (+ resid)
int s2st_bn_apply(const float* x, const float* mean, const float* var, const float* gamma,
                  const float* beta, float* y, Split ysp, const float* resid, int rows, int C,
                  float eps, int tanh_, float drop_p, uint64_t seed, hipStream_t st);
// the same transform as the next convolution's operand: bf16 halo image img [B][T + 2 pad][C] with zero halos, and
// (y != null) the fp32 result rows [B * T][C]
int s2st_bn_apply_img(const float* x, const float* mean, const float* var, const float* gamma, const float* beta, float* y,
                      uint16_t* img, int B, int T, int pad, int C, float eps, int tanh_, float drop_p, uint64_t seed,
                      hipStream_t st);
// dx (via dxsp) = BN/tanh/dropout backward; dgamma/dbeta += ; tmp = 2*C floats
int s2st_bn_bwd(const float* dy, Split dysp, const float* x, const float* mean, const float* var,
                const float* gamma, const float* beta, float* dx, Split dxsp, float* dgamma,
                float* dbeta, float* tmp, int rows, int C, float eps, int tanh_, float drop_p,
                uint64_t seed, hipStream_t st, uint16_t* dxh = nullptr /* optional bf16 twin [rows][ldh] */,
                long ldh = 0);

// ---------------------------------------------------------------------------------------
// fused attention (attention.hip)
// ---------------------------------------------------------------------------------------
int s2st_flash_attn_supported(int dh);
int s2st_flash_attn_preload(hipStream_t st);
int s2st_flash_attn_fwd(const s2st_attn_args* p, hipStream_t st);
int s2st_flash_attn_bwd(const s2st_attn_args* p, const float* dO, float* dvec_scratch, hipStream_t st, int phase = 0,
                        float* db_part = nullptr /* bias gradients as partial sums + ordered fold: s2st_flash_attn_db_scratch_floats */);
long s2st_flash_attn_db_scratch_floats(const s2st_attn_args* p);
// layout of that scratch: [slots_q][C] (q) | [slots_k][C] (k) | [slots_k][C] (v), C = H * dh
void s2st_flash_attn_db_layout(const s2st_attn_args* p, int* slots_q, int* slots_k);
int s2st_flash_attn_db_fold(const s2st_attn_args* p, const float* db_part, hipStream_t st);  // db += fold(partials), slot order

// ---------------------------------------------------------------------------------------
// HuBERT front end (hubert.hip): waveform conv, GroupNorm(C, C) + GELU, pos-conv input re-layout
// ---------------------------------------------------------------------------------------
int s2st_hubert_conv0_gn_gelu(const float* x, const float* w, const float* gamma, const float* beta, float* y,
                              uint16_t* yh, float* stats /* s2st_hubert_conv0_stats_floats(B, T, C) */, int B, int N, int T, int C, int k,
                              int stride, float eps, hipStream_t st);
long s2st_hubert_conv0_stats_floats(int B, int T, int C);
int s2st_posconv_prep(float* x, const int* lens, float* img, uint16_t* imgh, int B, int T, int E, int G, int pad,
                      int Tp, hipStream_t st);

// ---------------------------------------------------------------------------------------
// inference (infer.hip): incremental-decoding attention, stop / alignment / de-CMVN helpers, Griffin-Lim
// ---------------------------------------------------------------------------------------
int s2st_decode_attn(const float* q, long ldq, float* kc, float* vc, long ldk, long kbs, const int* klen,
                     int nkeys, int B, int H, int dh, float scale, float* o, long ldo, float* attn_mean, int S,
                     hipStream_t st, const float* k_new = nullptr, const float* v_new = nullptr, long ld_new = 0,
                     int pos_new = 0, int kv_bf16 = 0,  // kv_bf16: kc / vc point at bf16 rows (ldk, kbs in elements)
                     const int* step_ptr = nullptr);  // replayable decode step: nkeys = *step_ptr + 1, pos_new = *step_ptr (nkeys: the bound)
int s2st_scale_rows(const float* x, const float* a, float* y, long n, hipStream_t st);
int s2st_cache_reorder(const float* src, float* dst, const int* idx, int nb, int Bb, long row_floats, long valid_floats, hipStream_t st);
int s2st_decode_stop_update(const float* eos_prob, float thr, int step, int max_iter, int B, int* finished, int* out_lens,
                            int* klen_next, int* n_done, hipStream_t st);
// replayable decode step (engine.cpp decode_step replay mode): state for step 0; stop rule + output rows + next step's state
int s2st_decode_replay_init(int* step, uint64_t* seeds, float* cur_feat, long n_feat, float* pe_cur, const float* pe_alpha, int Cd,
                            uint64_t seed0, hipStream_t st);
int s2st_decode_replay_commit(int* step, uint64_t* seeds, const float* cur_feat, const float* cur_eos, const float* cur_attn,
                              float* pe_cur, const float* pe_alpha, int pe_rows, int Cd, uint64_t seed0, float thr, int max_iter,
                              int B, int out_dim, int E, int* finished, int* out_lens, int* klen_next, int* n_done,
                              float* feat_all, float* eos_all, float* attn_all, hipStream_t st);
int s2st_sigmoid(const float* x, float* y, long n, hipStream_t st);
int s2st_argmax_dim1(const float* x, long* idx, int B, int E, int D, hipStream_t st);
int s2st_affine_cols(const float* x, const float* scale, const float* shift, float* y, long rows, int C, hipStream_t st);
int s2st_exp_transpose(const float* x, float* y, int T, int C, hipStream_t st);
int s2st_clamp_min(float* x, long n, float lo, hipStream_t st);
int s2st_gl_polar(const float* mag, const float* ang, float* X, int F, int T, hipStream_t st);
int s2st_gl_project(const float* mag, const float* Y, float* X, int F, int T, hipStream_t st);
int s2st_reflect_pad(const float* x, float* y, int n, int pad, hipStream_t st);
int s2st_gl_polar_split(const float* mag, const float* aux, int from_spectrum, const int* tl, uint16_t* Xs, int U, int F,
                        int Fp, int Tmax, hipStream_t st);
int s2st_gl_frame_split(const float* wave, const int* tl, uint16_t* As, int U, int Tmax, int hop, int n_fft, int Lw,
                        hipStream_t st);
int s2st_gl_overlap_add_b(const float* frames, const float* wsq_all, const long* wsq_off, const int* tl, float* wave,
                          int U, int Tmax, int n_fft, int hop, int Lw, hipStream_t st);
// FFT-based Griffin-Lim (infer.hip): n_fft a power of two in 256 ... 2048; X is complex [U * Tmax][n_fft / 2 + 1] (re, im
// interleaved); win [n_fft]; tw [n_fft] complex = exp(-2 pi i j / n_fft)
bool s2st_gl_fft_supported(int n_fft);
int s2st_gl_polar_c(const float* mag, const float* ang, const int* tl, float* X, int U, int F, int Tmax, hipStream_t st);
// initial phases from uniform draws (uni: utterance u's [F][T_u] block at uni + uoff[u], doubles as numpy drew them) or, uni ==
// nullptr, from the device's counter-based generator
int s2st_exp_inplace(float* x, long n, hipStream_t st);
int s2st_gl_polar_u(const float* mag, const double* uni, const long* uoff, const int* tl, uint64_t seed, float* X, int U, int F,
                    int Tmax, hipStream_t st);
int s2st_gl_stft_project(const float* wave, const int* tl, const float* win, const float* tw, const float* mag, float* X, int U,
                         int Tmax, int n_fft, int hop, int Lw, hipStream_t st);
int s2st_gl_istft_ola(const float* X, const int* tl, const float* win, const float* tw, const float* wsq_all, const long* wsq_off,
                      float* wave, int U, int Tmax, int n_fft, int hop, int Lw, hipStream_t st);
int s2st_gl_istft_frames(const float* X, const int* tl, const float* win, const float* tw, float* frames, int U, int Tmax,
                         int n_fft, int hop, hipStream_t st);
int s2st_gl_overlap_add(const float* frames, const float* wsq, float* wave, int T, int n_fft, int hop, int n_out,
                        hipStream_t st);

// ---------------------------------------------------------------------------------------
// MCD evaluation (metrics.hip)
// ---------------------------------------------------------------------------------------
int s2st_dtw(const float* dist, const int* shapes, int B, int M, int N, float* cum, int* backptr, int* pathmap,
             hipStream_t st);
int s2st_rms_dist(const float* x1, const float* x2, float* out, int m, int n, int D, long ldo, hipStream_t st);
int s2st_power_spec(const float* Y, float* P, int T, int F, hipStream_t st);
int s2st_log_offset(float* x, long n, float eps, hipStream_t st);

// ---------------------------------------------------------------------------------------
// losses (losses.hip)
// ---------------------------------------------------------------------------------------
// stats (optional) += {sum|fo-t| + sum|fp-t|, sum(fo-t)^2 + sum(fp-t)^2, sum bce} over valid
// steps t < lens[b]; stop target = 1 at t == lens[b]-1.  Gradients (optional):
//   dfeat/dpost = c_l1 * sign(e) + c_mse * 2e ; deos = c_eos * dBCE/dx ; zero on padded steps
int s2st_mel_loss(const float* feat, const float* post, const float* eos, const float* tgt,
                  const int* lens, int B, int D, int F, float pos_weight, float* stats, float c_l1,
                  float c_mse, float c_eos, float* dfeat, float* dpost, float* deos,
                  hipStream_t st, float* ordered = nullptr, int* nblocks_out = nullptr);
// ordered (also s2st_ls_ce): scratch of S2ST_LOSS_ORDERED_FLOATS floats -- the workgroups' sums go there instead of into
// stats by float atomics ([k][nblocks], nblocks returned); s2st_loss_finalize(parts) adds them in workgroup order: the
// logged sums repeat bit for bit
#define S2ST_LOSS_ORDERED_FLOATS (4 * 2048)
struct s2st_loss_parts {  // per loss kernel (mel, ASR CE, ST CE): its scratch and workgroup count (null / 0: not run)
  const float* part[3];
  int nblocks[3];
  int on;
};
// label-smoothed CE over logits [rows][V]; stats (optional) += {nll_sum, smooth_sum,
// n_correct, total}; dlogits (optional) = gscale * d/dlogits[(1-eps-eps_i) nll + eps_i smooth]
int s2st_ls_ce(const float* logits, const long* target, int rows, int V, long pad, float eps,
               float* stats, float* dlogits, float gscale, hipStream_t st, float* ordered = nullptr,
               int* nblocks_out = nullptr);
// log_softmax + CTC (blank 0, zero_infinity).  logits [B][E][V]; targets [B][Lmax];
// lprobs [B][E][V] (required, also an output); loss_per_utt[b] = nll_b / max(L_b, 1);
// dlogits (optional) = gscale / max(L_b,1) * (softmax - occupancy), 0 for t >= in_lens[b]
long s2st_ctc_workspace_floats(int B, int E, int Lmax);
int s2st_ctc(const float* logits, const long* targets, int Lmax, const int* in_lens,
             const int* tgt_lens, int B, int E, int V, float* lprobs, float* loss_per_utt,
             float* dlogits, float gscale, float* ws, hipStream_t st);

int s2st_log_softmax_rows(const float* x, long ldx, float* y, long ldy, int rows, int V, int log_out, hipStream_t st);
int s2st_loss_finalize(float* stats, const float* ctc_per, int B, float nf, float nr, float w_l1,
                       float w_mse, float w_eos, float w_ctc, float w_asr, float w_st, float eps, int Vs,
                       int Vt, float src_ntok, float tgt_ntok, hipStream_t st, const float* ctc_tgt_per = nullptr,
                       float w_ctc_tgt = 0.f, const s2st_loss_parts* parts = nullptr);

// ---------------------------------------------------------------------------------------
// optimizer (optim.hip)
// ---------------------------------------------------------------------------------------
int s2st_sumsq(const float* x, long n, float* out /* += */, hipStream_t st);
#define S2ST_SUMSQ_PARTS 1024
// the same sum as s2st_sumsq_nparts(n) <= S2ST_SUMSQ_PARTS per-block partials: s2st_adam(sumsq_parts = count) adds
// them in index order -- run-to-run identical, no zeroing pass, no atomics
int s2st_sumsq_parts(const float* x, long n, float* parts, hipStream_t st);
long s2st_sumsq_nparts(long n);
// g *= gmul * clip ; clip = min(1, max_norm / (sqrt(sumsq)*gmul + 1e-6)) ; fairseq Adam.
// step >= 1 is the Adam step count; gnorm_out (optional) receives sqrt(sumsq)*gmul.
// effective gradient multiplier = gmul * (gmul_dev ? *gmul_dev : 1)
int s2st_adam(float* p, float* g, float* m, float* v, long n, const float* sumsq, float gmul,
              const float* gmul_dev, float max_norm, float lr, float beta1, float beta2, float eps, float wd, int step,
              float* gnorm_out, hipStream_t st, uint16_t* p_bf16 = nullptr /* optional bf16 copy of the new p */,
              int* skipped = nullptr /* optional device counter: += 1 when the norm is non-finite and the update is skipped */,
              int sumsq_parts = 0 /* > 0: sumsq points at that many partial sums (s2st_sumsq_parts) */,
              int zero_grad = 0 /* 1: g is left all zero (also when the update is skipped) instead of holding the scaled
                                   gradient: the next step needs no clearing pass over the arena */);
