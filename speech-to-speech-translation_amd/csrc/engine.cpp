// s2st training engine: forward + loss + backward of the s2st_transformer as one stream-ordered
// schedule of HIP kernels over flat parameter / gradient arenas and a bump-allocated
// activation workspace.  No tracing compiler, no autograd graph: the forward pushes one
// backward closure per op on a tape; the backward pops them.  Parameters are laid out in
// forward-use order so the backward finishes gradient ranges back-to-front, which lets the
// host overlap RCCL all-reduce of finished ranges with the rest of the backward.
//
// Reference behaviour reproduced (file:line under /root/reference):
//   encoder   examples/s2s_trans/models/s2st_transformer.py:94-140, 195-237
//   decoder   s2st_transformer.py:369-456 ; Prenet/Postnet fairseq/models/text_to_speech/tacotron2.py:85-126
//   layers    fairseq/modules/transformer_layer.py:107-165, 301-446 ; MHA multihead_attention.py:160-385
//   aux text decoders  s2st_transformer.py:483-578 ; transformer_decoder.py:253-378
//   criterion examples/s2s_trans/criterions/s2st_loss.py:179-315
// Layout: activations are [B][T][C] row-major (the reference is [T][B][C]); padded rows are
// computed exactly as the reference computes them.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "s2st_ops.h"
#include <hip/hip_ext.h>

namespace {

struct PInfo {
  std::string name;
  long off, numel;
  int ndim;
  int shape[4];
  int is_buffer;
};

struct Ten {  // plain [rows][cols] fp32 activation
  float* d = nullptr;
  float* g = nullptr;
  // fast (bf16-operand) mode: bf16 copies of d / g with row stride hld(), made by the producer
  // kernel when it can, else by a cast kernel at the first GEMM that reads them
  bf16raw* h = nullptr;
  bf16raw* gh = nullptr;
  int rows = 0, cols = 0;
  bool needs_grad = true;
  bool want_gh = false;  // the gradient is consumed as a GEMM operand: its producer writes gh too
  // produced by linear(act = ReLU, dropout p) with a single consumer (FFN hidden): the consumer's
  // data-gradient GEMM applies the activation backward + bias gradient in its epilogue and leaves the
  // ready bf16 operand in gpre_h (no fp32 gradient of this tensor is ever written)
  int act_mode = 0;
  float act_p = 0.f;
  long act_bias = -1;
  // produced by linear(no activation, dropout p) [+ residual]: the layer norm that consumes it first (hence last
  // in the backward, when its gradient is complete) emits the bf16 operand dropout'(g) + bias gradient (gpre_h)
  bool drop2_ok = false, ln_seen = false;
  float drop2_p = 0.f;
  uint64_t drop2_seed = 0;
  long drop2_bias = -1;
  bool lin_plain = false;  // output of a bias-only linear (no activation / dropout / residual): its gradient IS the GEMM operand
  bf16raw* gpre_h = nullptr;
  long n() const { return (long)rows * cols; }
  int hld() const { return (cols + 7) & ~7; }
};

struct LinP { long w, b; int N, K; };           // offsets into the param arena (b < 0: no bias)
struct LNP { long g, b; int C; };
struct AttnP { long kvq_w, kvq_b, out_w, out_b; };                 // self: [3C][C] k,v,q rows
struct XAttnP { long kv_w, kv_b, q_w, q_b, out_w, out_b; };         // cross: kv [2C][Cenc]
struct EncLayerP { AttnP sa; LNP ln1; LinP fc1, fc2; LNP ln2; };
struct DecLayerP { AttnP sa; LNP ln1; XAttnP xa; LNP ln2; LinP fc1, fc2; LNP ln3; };
struct ConvP { long w, b; int O, I, Kw; };
struct BNP { long g, b, rm, rv; int C; };
struct AuxP {
  long embed; int V, in_dim, d, layers, out_dim;
  long proj_in;  // -1 if none
  std::vector<DecLayerP> L;
  LNP ln; bool has_ln;
  long proj_out; // -1 if none
  long out_proj;
};

}  // namespace

struct s2st_engine {
  s2st_model_config c;
  std::vector<PInfo> infos;
  long n_params = 0, n_buffers = 0;
  float *P = nullptr, *G = nullptr, *BUF = nullptr;
  // weight-gradient GEMMs are off the backward critical path: they run on a second stream, next
  // to the data-gradient chain (each of these GEMMs alone fills about half of the 256 CUs)
  hipStream_t side_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr, ev_taps_ = nullptr;
  bool overlap_aux = true;  // S2ST_NO_AUX_OVERLAP=1 (A/B switch)
  bool hoist_kv = true;     // S2ST_NO_KV_HOIST=1 (A/B switch): cross-attention K|V projections inside the layers
  hipEvent_t ev_kv_ = nullptr;
  bool kv_wait_ = false;    // the data path has not yet waited for the hoisted K|V projections
  hipEvent_t ev_auxb_ = nullptr;
  size_t aux_wait_idx = 0, aux_lo_idx = 0, aux_hi_idx = 0;  // tape indices: tap-LN end; aux section [lo, hi)
  bool aux_bwd_on_side = false;
  bool side_used = false;
  float* skws_side = nullptr;
  bool side_allowed = false;  // bf16-operand mode and no S2ST_NO_SIDE_STREAM=1
  bool side_tried = false;
  void ensure_side() {        // the second stream and its events, on first need (the first training forward; the overlapped update)
    if (side_ || side_tried || !side_allowed) return;
    side_tried = true;
    // the side stream carries weight gradients nobody waits for until the segment ends: lowest queue priority, so that
    // when both queues have workgroups ready the data path (the critical chain of the backward) is dispatched first
    int pr_least = 0, pr_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest) != hipSuccess) pr_least = 0;
    if (hipStreamCreateWithPriority(&side_, hipStreamNonBlocking, pr_least) != hipSuccess) side_ = nullptr;
    if (side_ && hipEventCreateWithFlags(&ev_kv_, hipEventDisableTiming) != hipSuccess) ev_kv_ = nullptr;
    if (side_ && (hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming) != hipSuccess ||
                  hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming) != hipSuccess ||
                  hipEventCreateWithFlags(&ev_taps_, hipEventDisableTiming) != hipSuccess ||
                  hipEventCreateWithFlags(&ev_auxb_, hipEventDisableTiming) != hipSuccess)) {
      hipStreamDestroy(side_);
      side_ = nullptr;
    }
  }
  hipStream_t fork_side() {  // everything issued on st_ so far happens-before what follows on the returned stream
    if (!side_) { sync_chains(); return st_; }
    hipEventRecord(ev_fork_, st_);
    hipStreamWaitEvent(side_, ev_fork_, 0);
    if (forked_) {  // ... and everything issued on the second chain
      hipEventRecord(ev_cside_, chain1_);
      hipStreamWaitEvent(side_, ev_cside_, 0);
    }
    side_used = true;
    return side_;
  }
  // S2ST_STALL_TRACE=1: time spent by the data-path stream in each cross-stream wait (timing events around the
  // wait; read back by s2st_engine_stall_report after the caller synchronised)
  struct Stall { const char* what; hipEvent_t a, b; };
  std::vector<Stall> stalls;
  bool stall_trace = false;
  void wait_traced(hipStream_t st, hipEvent_t ev, const char* what) {
    if (!stall_trace) { hipStreamWaitEvent(st, ev, 0); return; }
    Stall s{what, nullptr, nullptr};
    hipEventCreate(&s.a);
    hipEventCreate(&s.b);
    hipEventRecord(s.a, st);
    hipStreamWaitEvent(st, ev, 0);
    hipEventRecord(s.b, st);
    stalls.push_back(s);
  }
  void join_side() {
    sync_chains();
    if (!side_ || !side_used) return;
    hipEventRecord(ev_join_, side_);
    wait_traced(st_, ev_join_, "join_side");
    side_used = false;
  }

  // ---- two utterance-half chains (round 5; S2ST_CHAINS=2) -----------------------------------------
  // Between the conv front end and the post-net every op of the training step is independent across utterances: rows of
  // the linear layers and layer norms, attention per (utterance, head).  In this mode those ops are launched TWICE -- rows
  // of utterances [0, B/2) on the caller's stream, rows of [B/2, B) on a second chain stream -- over the SAME whole-batch
  // tensors, so that one chain's per-kernel latency (launch boundary, prologue, a single round of tiles) runs under the
  // other's.  Everything that spans the batch stays ONE launch behind both chains: the weight-gradient products (K = all
  // tokens: fork_side waits for both), BatchNorm / convolutions / losses (sync_chains in front of them), the partial-sum
  // folds (both chains' partial rows, chain 0's first).  Dropout masks are keyed by (seed, element index of the LAUNCH):
  // chain 1 salts its seed, so its masks are independent of chain 0's (they differ from the one-chain schedule's masks for
  // those rows -- same distribution; with dropout off the two schedules' forward outputs are bit-identical).
  int nchains = 1;               // S2ST_CHAINS
  hipStream_t chain1_ = nullptr;
  hipEvent_t ev_cfork_ = nullptr, ev_cjoin_ = nullptr, ev_cside_ = nullptr;
  bool forked_ = false;
  bool in_region_ = false;       // forward: between chain_region(true) and chain_region(false); backward: per closure
  hipStream_t main_ = nullptr;   // the caller's stream of this call (st_ points at side_ while the aux sections run)
  float* skws_c1 = nullptr;      // split-K scratch of the second chain
  std::vector<char> tape_aware;  // closure i launches per chain itself (everything else gets sync_chains() first)
  static constexpr uint64_t CHAIN_SALT = 0xD6E8FEB86659FD93ULL;
  struct Part { int r0, nr; hipStream_t st; uint64_t salt; };
  // how many parts an op over `rows` rows (rows = B x positions) is launched in -- independent of live(), so that the dry
  // run of the schedule allocates what the real one does
  int chain_count(int rows) const {
    return (nchains == 2 && chain1_ && in_region_ && st_ == main_ && bt.B >= 2 && rows > 0 && rows % bt.B == 0) ? 2 : 1;
  }
  int chain_parts(int rows, Part* p) {
    if (chain_count(rows) == 1) {
      if (live()) sync_chains();
      p[0] = Part{0, rows, st_, 0};
      return 1;
    }
    const int per = rows / bt.B, b0 = bt.B / 2;
    if (live()) ensure_forked();
    p[0] = Part{0, b0 * per, main_, 0};
    p[1] = Part{b0 * per, rows - b0 * per, chain1_, CHAIN_SALT};
    return 2;
  }
  void ensure_forked() {
    if (forked_) return;
    hipEventRecord(ev_cfork_, main_);
    hipStreamWaitEvent(chain1_, ev_cfork_, 0);
    forked_ = true;
  }
  void sync_chains() {  // the caller's stream takes in what the second chain did
    if (!forked_) return;
    hipEventRecord(ev_cjoin_, chain1_);
    hipStreamWaitEvent(main_, ev_cjoin_, 0);
    forked_ = false;
  }
  void set_aware() {
    tape_aware.resize(tape.size(), 0);
    if (!tape.empty()) tape_aware.back() = in_region_ ? 1 : 0;
  }
  void chains_wait(hipEvent_t ev) {  // an event both chains have to honour (work handed over from the second stream)
    if (forked_) hipStreamWaitEvent(chain1_, ev, 0);
  }
  // ---- AR decoding state (config 5): caller-owned cache buffer laid out by decode_begin --------
  // replay_ != null: decode_step in its graph-replayable form (s2st_engine_decode_step_replay): the step, the prenet's dropout
  // seeds, the input frame, the position row and the output rows are read from / written to the fixed device buffers of *replay_
  const s2st_decode_replay* replay_ = nullptr;
  // merged decode batches (s2st_engine_decode_row_map): device [B] -- the row of its OWN batch whose Prenet dropout mask row b draws
  const int* dec_row_map = nullptr;
  bool stop_after_encoder = false;
  Ten* enc_out_keep = nullptr;
  struct Dec {
    float* base = nullptr; int B = 0, E = 0, maxT = 0; const int* enc_lens = nullptr; const float* pe_dec = nullptr;
    float* pe_alpha = nullptr;  // [maxT + 2][dec_dim]: pos_emb_alpha * PE rows (tail of the caller's state buffer)
  } dec_st;
  float* dec_selfK(int l) const { return dec_st.base + (long)l * 2 * dec_st.B * dec_st.maxT * c.dec_dim; }
  float* dec_selfV(int l) const { return dec_selfK(l) + (long)dec_st.B * dec_st.maxT * c.dec_dim; }
  float* dec_crossKV(int l) const {
    return dec_st.base + (long)c.dec_layers * 2 * dec_st.B * dec_st.maxT * c.dec_dim + (long)l * dec_st.B * dec_st.E * 2 * c.dec_dim;
  }
                             // (0 = none).  Balances the two streams; measured on the bench workload: n = 3 .. 16,
                             // best 7 (11.15 -> 10.89 ms/step together with the attention-backward bf16 gradients)
  bool use_ln_skinny = true; // S2ST_NO_LN_SKINNY=1 (A/B switch): separate layer-norm kernels in the AR decoding steps
  bool use_skinny = true;    // S2ST_NO_SKINNY=1 (A/B switch): tiled GEMMs for the AR decoding steps too
  bool use_ln_fuse = true;  // S2ST_NO_LN_FUSE=1 (A/B switch): separate dropout-backward prologue pass
  bool use_only_h = true;  // S2ST_NO_ONLY_H=1: always keep the fp32 copy of GEMM-only tensors (A/B switch)
  // S2ST_ATTN_GFUSE (default 1): the attention backward emits the bf16 projection gradients itself (no fp32 gradient, no
  // cast pass: ~30 launches fewer on the data path); the modes differ in where the projections' bias gradients come from
  // (attention block below).  1 (default since round 3): out of the kernels' fp32 accumulators BEFORE rounding, as per-(block,
  // wave) partial sums by DPP row reductions, folded in a fixed order with the segment's layer-norm partials -- exact (a key
  // bias's mathematically zero gradient stays ~0) and run-to-run identical; 2: the same sums as fp32 atomics; 3: column sums of
  // the ROUNDED copies (round 1's form: a rounding residue instead of ~0 that changes with any upstream ulp -- multi-update
  // trajectories of cold processes then repeat in ~85 % of runs); 0: off (fp32 gradients + cast pass).
  bool use_attn_gfuse = true;
  int attn_gfuse_mode = 1;
  // Ordered sums (default since round 3; S2ST_ORDERED_BIAS_SUMS=0 restores the atomics, an A/B switch): every sum that
  // used fp32 atomics -- bias-gradient column sums (linear backward prologue, conv biases, the masked data-gradient
  // GEMM epilogue), embedding-row gradients, the decoder's position scale -- is formed as partial rows + a fold in index
  // order, so a gradient is a function of (parameters, batch, seed) down to the last bit.  Round 2 had this as an
  // option at +3 % of a step (one small fold launch per sum); the folds now ride in the segment's batched fold launch
  // (pending_lnfold), i.e. cost no launches of their own.
  bool ordered_sums = true;
                               // Emitting the bf16 GEMM operands from the attention backward takes 0.6 ms off the
                               // data-path stream but makes the second stream the longer one (its final join grew from
                               // 0.1 to 0.6 ms): it only pays together with wgrad_main_every below
  bool use_act_fuse = true;  // S2ST_NO_ACT_FUSE=1: separate ReLU-dropout backward kernel (A/B switch)
  bool use_flash = true;  // S2ST_NO_FLASH=1: unfused attention everywhere (A/B switch)
  int ffn_act = 1;        // 1 relu (s2st layers), 2 gelu (HuBERT layers)
  // ---- frozen HuBERT front end (config 4): same engine object in "hubert mode" -------------
  bool is_hubert = false;
  s2st_hubert_config hc{};
  struct HubP {
    long conv_w[8]; long gn_g, gn_b; LNP ln; LinP proj; long pos_w, pos_b; std::vector<EncLayerP> L; LNP enc_ln;
  } hp;
  float* ws_for(hipStream_t s) const { return (side_ && s == side_) ? skws_side : ((chain1_ && s == chain1_) ? skws_c1 : skws); }
  float* skws = nullptr;  // split-K partial-sum scratch of the weight-gradient GEMMs (per call)
  long skws_n = 0;
  bool ph_fresh = false;   // s2st_engine_bf16_is_fresh: PH already equals bf16(P) for the next forward
  bf16raw* PHT = nullptr;  // optional: transposed bf16 copies of the 2-D weights (same offsets): the
                           // data-gradient GEMMs then read K-contiguous operands (~25 % faster here)
  bool pht_valid = false;
  struct WT { long off; int N, K; };
  std::vector<WT> wt_list;  // the [N][K] matrices linear() multiplies by (fused k|v|q blocks as ONE matrix)
  void reg_wt(long off, int N, int K) { if (N % 8 == 0 && K % 8 == 0) wt_list.push_back(WT{off, N, K}); }
  bool has_wt(long off, int N, int K) const {
    if (!pht_valid) return false;
    for (const WT& t : wt_list) if (t.off == off && t.N == N && t.K == K) return true;
    return false;
  }
  std::vector<s2st_transpose_table> wt_tables;
  void build_wt_tables() {
    s2st_transpose_table cur{};
    for (const WT& t : wt_list) {
      if (cur.n == S2ST_TRANSPOSE_MAX) { wt_tables.push_back(cur); cur = s2st_transpose_table{}; }
      const int i = cur.n++;
      cur.off[i] = (unsigned)t.off; cur.rows8[i] = (unsigned short)(t.N / 8); cur.cols8[i] = (unsigned short)(t.K / 8);
      cur.tile0[i + 1] = cur.tile0[i] + (unsigned)(((t.N + 63) / 64) * ((t.K + 63) / 64));
    }
    if (cur.n) wt_tables.push_back(cur);
  }
  bf16raw* PH = nullptr;  // bf16 copy of the parameter arena (same offsets), refreshed every forward
  bool f32_operands = false;  // debug A/B switch S2ST_F32_OPERANDS=1: bf16 MFMA on fp32-stored operands
  bool fast() const { return c.precise == 0 && !f32_operands; }

  // parameter handles
  ConvP sub[2];
  std::vector<EncLayerP> enc;
  LNP enc_ln; bool has_enc_ln = false;
  LNP asr_norm, st_norm;
  long pos_alpha = 0;
  std::vector<LinP> prenet;   // prenet_layers + 1
  std::vector<DecLayerP> dec;
  LNP dec_ln; bool has_dec_ln = false;
  LinP feat_proj, eos_proj;
  std::vector<ConvP> post_conv;
  std::vector<BNP> post_bn;
  LinP ctc_proj, ctc_proj_tgt;
  AuxP asr, st;
  AuxP s2t;  // s2t_mode: the model's own text decoder ("decoder.*")

  // per-call state
  float* ws = nullptr;
  long ws_cap = 0, ws_top = 0, ws_peak = 0;
  bool dry = false, oom = false;
  int err = 0;
  hipStream_t st_ = nullptr;
  std::vector<std::function<void()>> tape;
  std::vector<Ten*> tens;
  struct Mark { size_t tape_idx; long param_off; };
  std::vector<Mark> marks;
  long param_watermark = 0;
  int next_segment = 0;
  uint64_t seed = 0, site = 0;
  float gscale = 1.f;
  s2st_batch bt;
  s2st_outputs outs;

#include "engine_params.h"  // parameter construction

#include "engine_runtime.h"  // per-call runtime

#include "engine_ops.h"  // the ops of the training graph

#include "engine_decode.h"  // incremental decoding

#include "engine_convnets.h"  // the convolutional sub-networks

#include "engine_hubert.h"  // the frozen HuBERT front end (config 4)

#include "engine_step.h"  // one step

};

// ---------------------------------------------------------------------------------------------
extern "C" {

int s2st_engine_create(const s2st_model_config* cfg, s2st_engine** out) {
  if (!cfg || !out) return S2ST_ERR_ARG;
  if (cfg->enc_dim % cfg->enc_heads || cfg->dec_dim % cfg->dec_heads || cfg->enc_dim % 4 ||
      cfg->dec_dim % 4 || cfg->in_dim % 4 || cfg->out_dim % 4 || cfg->conv_channels % 8)
    return S2ST_ERR_SHAPE;
  // bf16-operand mode reads 16-byte chunks of 8 elements: every projection width must be % 8
  if (!cfg->precise && (cfg->enc_dim % 8 || cfg->dec_dim % 8 || cfg->in_dim % 8 || cfg->out_dim % 8 ||
                        cfg->prenet_dim % 8 || cfg->postnet_dim % 8 || cfg->enc_ffn % 8 || cfg->dec_ffn % 8 ||
                        cfg->conv_channels % 16 || (cfg->has_asr && cfg->asr_dim % 8) ||
                        (cfg->has_st && cfg->st_dim % 8)))
    return S2ST_ERR_SHAPE;
  s2st_engine* e = new s2st_engine();
  e->c = *cfg;
  e->f32_operands = s2st_env_on("S2ST_F32_OPERANDS");
  e->use_flash = !(s2st_env_on("S2ST_NO_FLASH"));
  e->group_wgrad = !(s2st_env_on("S2ST_NO_WGRAD_GROUP"));
  if (s2st_env_str("S2ST_WGRAD_GROUP")) {
    e->group_flush_at = atoi(s2st_env_str("S2ST_WGRAD_GROUP"));
    if (e->group_flush_at < 1) e->group_flush_at = 1;
    if (e->group_flush_at > S2ST_GROUP_MAX) e->group_flush_at = S2ST_GROUP_MAX;
  }
  e->use_act_fuse = !(s2st_env_on("S2ST_NO_ACT_FUSE"));
  e->use_only_h = !(s2st_env_on("S2ST_NO_ONLY_H"));
  e->hoist_kv = !(s2st_env_on("S2ST_NO_KV_HOIST"));
  e->stall_trace = s2st_env_on("S2ST_STALL_TRACE");
  e->use_ln_skinny = !(s2st_env_on("S2ST_NO_LN_SKINNY"));
  e->use_skinny = !(s2st_env_on("S2ST_NO_SKINNY"));
  e->use_ln_fuse = !(s2st_env_on("S2ST_NO_LN_FUSE"));
  e->attn_gfuse_mode = s2st_env_int("S2ST_ATTN_GFUSE", 1);
  e->use_attn_gfuse = e->attn_gfuse_mode != 0;
  e->ln_bwd_split = s2st_env_on("S2ST_LN_BWD_SPLIT");
  e->ordered_sums = !(s2st_env_int("S2ST_ORDERED_BIAS_SUMS", 1) == 0);
  e->build_params();
  // (a process that replays HIP graphs can carry a stale "last error" of the runtime's own capture-time queries: the preload's
  //  launch checks must see their own errors only)
  (void)hipGetLastError();
  if (!cfg->precise && (s2st_gemm_bf16_preload(nullptr) != 0 || s2st_flash_attn_preload(nullptr) != 0)) { delete e; return S2ST_ERR_LAUNCH; }
  // The second stream is made at the first TRAINING forward (ensure_side), not here: a process has four hardware queues
  // (runtime/streams.py) and an engine that only ever decodes would hold one of them for nothing -- the caller's stream
  // then shares a queue with one of the generator's chains (config 5, profiles/r05_queue_matrix.txt).
  e->side_allowed = !cfg->precise && !(s2st_env_on("S2ST_NO_SIDE_STREAM"));
  e->overlap_aux = !(s2st_env_on("S2ST_NO_AUX_OVERLAP"));
  // S2ST_CHAINS=2: the training step's layers as two utterance-half chains (see chain_count)
  if (!cfg->precise && s2st_env_int("S2ST_CHAINS", 1) == 2) {
    if (hipStreamCreateWithFlags(&e->chain1_, hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&e->ev_cfork_, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&e->ev_cjoin_, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&e->ev_cside_, hipEventDisableTiming) == hipSuccess)
      e->nchains = 2;
    else { delete e; return S2ST_ERR_LAUNCH; }  // (asked for and not available: loud)
  }
  *out = e;
  return 0;
}

void s2st_engine_destroy(s2st_engine* e) {
  if (!e) return;
  if (e->side_) {
    hipStreamSynchronize(e->side_);
    hipStreamDestroy(e->side_);
    hipEventDestroy(e->ev_fork_);
    hipEventDestroy(e->ev_join_);
    if (e->ev_taps_) hipEventDestroy(e->ev_taps_);
    if (e->ev_auxb_) hipEventDestroy(e->ev_auxb_);
    if (e->ev_kv_) hipEventDestroy(e->ev_kv_);
    for (hipEvent_t ev : e->adam_ev) hipEventDestroy(ev);
  }
  if (e->chain1_) {
    hipStreamSynchronize(e->chain1_);
    hipStreamDestroy(e->chain1_);
    if (e->ev_cfork_) hipEventDestroy(e->ev_cfork_);
    if (e->ev_cjoin_) hipEventDestroy(e->ev_cjoin_);
    if (e->ev_cside_) hipEventDestroy(e->ev_cside_);
  }
  e->reset_call();
  delete e;
}

int32_t s2st_engine_num_params(const s2st_engine* e) { return (int32_t)e->infos.size(); }

int s2st_engine_param_info(const s2st_engine* e, int32_t i, s2st_param_info* out) {
  if (i < 0 || i >= (int)e->infos.size()) return S2ST_ERR_ARG;
  const PInfo& p = e->infos[i];
  memset(out, 0, sizeof(*out));
  strncpy(out->name, p.name.c_str(), sizeof(out->name) - 1);
  out->offset = p.off;
  out->numel = p.numel;
  out->ndim = p.ndim;
  for (int k = 0; k < 4; ++k) out->shape[k] = p.shape[k];
  out->is_buffer = p.is_buffer;
  return 0;
}

int64_t s2st_engine_param_floats(const s2st_engine* e) { return e->n_params; }
int64_t s2st_engine_buffer_floats(const s2st_engine* e) { return e->n_buffers; }

int s2st_engine_bind(s2st_engine* e, float* params, float* grads, float* buffers) {
  e->P = params;
  e->G = grads;
  e->BUF = buffers;
  return 0;
}

int s2st_engine_bind_bf16(s2st_engine* e, uint16_t* params_bf16) {
  e->PH = params_bf16;
  return 0;
}

// allow = 0: this engine never makes (or sizes scratch for) a second stream -- inference twins, which only ever decode
// (a stream created all the same would take a hardware-queue slot: runtime/streams.py).  Call before the first forward.
int s2st_engine_allow_side_stream(s2st_engine* e, int32_t allow) {
  if (!e) return S2ST_ERR_ARG;
  if (e->side_) return allow ? S2ST_OK : S2ST_ERR_ARG;  // (already made: cannot be taken back)
  e->side_allowed = allow != 0 && !e->c.precise;
  return S2ST_OK;
}

// Dropout-site log (test instrumentation, see Engine::next_seed): on = 1 makes every later forward record its sites;
// s2st_engine_site_log_get copies the records of the LAST forward (at most cap) and returns their number.
int s2st_engine_site_log(s2st_engine* e, int32_t on) {
  if (!e) return S2ST_ERR_ARG;
  e->site_log_on = on != 0;
  if (!on) e->site_log.clear();
  return S2ST_OK;
}
int32_t s2st_engine_site_log_get(const s2st_engine* e, s2st_dropout_site* out, int32_t cap) {
  if (!e) return -1;
  const int32_t n = (int32_t)e->site_log.size();
  for (int32_t i = 0; i < n && i < cap && out; ++i) out[i] = e->site_log[i];
  return n;
}

// S2ST_STALL_TRACE=1: prints (stderr) and clears the cross-stream waits recorded since the last report; the caller
// has synchronised the device.  Returns the total wait in microseconds.
int64_t s2st_engine_stall_report(s2st_engine* e, int32_t verbose) {
  if (!e) return S2ST_ERR_ARG;
  double total = 0;
  for (auto& s : e->stalls) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      total += ms * 1e3;
      if (verbose) fprintf(stderr, "[stall] %-44s %8.1f us\n", s.what, ms * 1e3);
    }
    hipEventDestroy(s.a);
    hipEventDestroy(s.b);
  }
  e->stalls.clear();
  return (int64_t)total;
}

int s2st_engine_bf16_is_fresh(s2st_engine* e) {
  if (!e || !e->PH) return S2ST_ERR_ARG;
  e->ph_fresh = true;
  return 0;
}

int s2st_engine_bind_bf16_transposed(s2st_engine* e, uint16_t* params_bf16_t) {
  e->PHT = params_bf16_t;
  return 0;
}

int64_t s2st_engine_workspace_floats(s2st_engine* e, const s2st_batch* b) {
  e->reset_call();
  e->bt = *b;
  memset(&e->outs, 0, sizeof(e->outs));
  e->dry = true;
  e->ws = reinterpret_cast<float*>(0x10000);  // never dereferenced: dry mode launches nothing
  e->ws_cap = (long)1 << 50;
  e->st_ = nullptr;
  // outputs the caller may keep internal are counted as workspace
  int rc = e->forward();
  if (rc == 0) {
    e->gscale = 1.f;
    for (int s = 0; s < e->n_segments(); ++s) e->backward_segment(s);
  }
  long peak = e->ws_peak;
  e->reset_call();
  e->dry = false;
  return rc ? (int64_t)rc : (int64_t)peak + 1024;
}

int s2st_engine_forward(s2st_engine* e, const s2st_batch* b, const s2st_outputs* out, float* workspace,
                        int64_t workspace_floats, void* stream) {
  if (!e->P || !e->BUF) return S2ST_ERR_ARG;
  e->reset_call();
  e->bt = *b;
  e->outs = *out;
  e->dry = false;
  e->ws = workspace;
  e->ws_cap = workspace_floats;
  e->st_ = (hipStream_t)stream;
  if (!e->outs.stats && b->tgt) return S2ST_ERR_ARG;
  return e->forward();
}

int s2st_engine_backward(s2st_engine* e, float gscale, int32_t segment, void* stream) {
  if (!e->G) return S2ST_ERR_ARG;
  e->st_ = (hipStream_t)stream;
  e->gscale = gscale;
  if (segment < 0) {
    for (int s = 0; s < e->n_segments(); ++s) {
      int rc = e->backward_segment(s);
      if (rc) return rc;
    }
    return 0;
  }
  return e->backward_segment(segment);
}

int s2st_engine_adam_overlapped(s2st_engine* e, float* exp_avg, float* exp_avg_sq, const float* sumsq_parts, int32_t n_parts,
                                float gmul, const float* gmul_dev, float max_norm, float lr, float beta1, float beta2, float eps,
                                float wd, int32_t step, float* gnorm_out, int32_t* skipped, int32_t write_bf16, int32_t n_chunks,
                                void* stream) {
  if (!e || !exp_avg || !exp_avg_sq || !sumsq_parts || n_parts <= 0) return S2ST_ERR_ARG;
  return e->adam_overlapped(exp_avg, exp_avg_sq, sumsq_parts, n_parts, gmul, gmul_dev, max_norm, lr, beta1, beta2, eps, wd, step,
                            gnorm_out, skipped, write_bf16, n_chunks, (hipStream_t)stream);
}

int s2st_engine_wait_optimizer(s2st_engine* e, void* stream) {
  if (!e) return S2ST_ERR_ARG;
  e->adam_wait_all((hipStream_t)stream);
  return 0;
}

int32_t s2st_engine_num_segments(const s2st_engine* e) { return e->n_segments(); }

void* s2st_engine_side_stream(const s2st_engine* e) {  // (a trainer asks before the first forward: made on demand)
  const_cast<s2st_engine*>(e)->ensure_side();
  return (void*)e->side_;
}

int s2st_engine_segment_range(const s2st_engine* e, int32_t i, int64_t* lo, int64_t* hi) {
  int ns = e->n_segments();
  if (i < 0 || i >= ns) return S2ST_ERR_ARG;
  *lo = e->marks[ns - i - 1].param_off;
  *hi = e->marks[ns - i].param_off;
  return 0;
}

// ---- AR decoding + eval post-net (config 5) ---------------------------------------------------
int64_t s2st_engine_decode_state_floats(const s2st_engine* e, int32_t B, int32_t E, int32_t max_steps) {
  const long Cd = e->c.dec_dim, L = e->c.dec_layers;
  // self-attention caches | static cross-attention keys and values | pad | alpha-scaled position rows | bf16 copies of the
  // cross-attention rows (fast mode)
  return L * 2 * B * (long)max_steps * Cd + L * (long)B * E * 2 * Cd + 64 + ((long)max_steps + 2) * Cd + L * (long)B * E * Cd + 8;
}

int s2st_engine_decode_begin(s2st_engine* e, const s2st_batch* b, const s2st_outputs* out, float* state,
                             int64_t state_floats, int32_t max_steps, float* workspace, int64_t workspace_floats,
                             void* stream) {
  if (!e->P || !e->BUF || !state || !b || !out) return S2ST_ERR_ARG;
  if (state_floats < s2st_engine_decode_state_floats(e, b->B, b->E, max_steps)) return S2ST_ERR_WORKSPACE;
  e->dec_row_map = nullptr;
  e->reset_call();
  e->bt = *b;
  e->bt.training = 0;
  e->bt.tgt = nullptr;
  e->outs = *out;
  e->dry = false;
  e->ws = workspace;
  e->ws_cap = workspace_floats;
  e->st_ = (hipStream_t)stream;
  e->stop_after_encoder = true;
  int rc = e->forward();
  e->stop_after_encoder = false;
  if (rc) return rc;
  e->dec_st.base = state;
  e->dec_st.B = b->B; e->dec_st.E = b->E; e->dec_st.maxT = max_steps;
  e->dec_st.enc_lens = b->enc_lens;
  e->dec_st.pe_dec = b->pe_dec;
  // alpha-scaled position rows 0 .. max_steps + 1 (one small launch per utterance batch instead of one per step)
  {
    const long Cd = e->c.dec_dim, L = e->c.dec_layers;
    e->dec_st.pe_alpha = state + L * 2 * b->B * (long)max_steps * Cd + L * (long)b->B * b->E * 2 * Cd + 64;
    e->touch(e->pos_alpha + 1);
    if (e->live())
      e->chk(s2st_scale_rows(b->pe_dec, e->P + e->pos_alpha, e->dec_st.pe_alpha, ((long)max_steps + 2) * Cd, e->st_));
  }
  // static cross-attention keys / values of every decoder layer (static_kv=True)
  // (bf16 copies of the static rows for the decode steps' cross-attention were built in round 4 and measured SLOWER than the
  //  fp32 rows -- 15.0 against 13.3 us per launch, profiles/r04_t_decode_attn_bench.txt; the kernel form stays in the C ABI
  //  (s2st_decode_attn, kv_bf16), the engine switch is gone)
  for (int l = 0; l < e->c.dec_layers; ++l) {
    const XAttnP& xa = e->dec[l].xa;
    e->linear(e->enc_out_keep, xa.kv_w, xa.kv_b, 2 * e->c.dec_dim, e->c.enc_dim, 0, 0.f, nullptr, e->dec_crossKV(l));
  }
  e->tape.clear();
  return e->err;
}

// Several batches decoded as ONE (round 6): the always-on Prenet dropout (tacotron2.py:95-98) keys its mask by the row of the
// batch, so the rows of the 2nd, 3rd ... batch say which row of their own batch they are.  NULL: rows count from 0.
// Set after decode_begin (which clears it), holds for the run's steps; the non-skinny paths (more than 256 rows) ignore it.
int s2st_engine_decode_row_map(s2st_engine* e, const int32_t* row_map) {
  if (!e) return S2ST_ERR_ARG;
  e->dec_row_map = row_map;
  return S2ST_OK;
}

int s2st_engine_decode_step(s2st_engine* e, int32_t step, const float* prev, const int32_t* pos,
                            const int32_t* self_klen, uint64_t seed, float* feat_out, float* eos_prob,
                            float* attn_out, float* workspace, int64_t workspace_floats, void* stream) {
  if (!e->P || !prev || !pos || !feat_out || !eos_prob) return S2ST_ERR_ARG;
  s2st_engine::Dec keep = e->dec_st;
  s2st_batch bt_keep = e->bt;
  e->reset_call();
  e->dec_st = keep;
  e->bt = bt_keep;
  e->dry = false;
  e->ws = workspace;
  e->ws_cap = workspace_floats;
  e->st_ = (hipStream_t)stream;
  e->skws = nullptr; e->skws_n = 0; e->skws_side = nullptr;
  if (e->fast()) {
    if (!e->PH) return S2ST_ERR_ARG;  // PH was refreshed by decode_begin's forward
  }
  int rc = e->decode_step(step, prev, pos, self_klen, seed, feat_out, eos_prob, attn_out);
  e->tape.clear();
  return rc;
}

// ---- the decode step in its graph-replayable form (include/s2st_hip.h: s2st_decode_replay) --------------------------------
// 1 when the run decode_begin prepared is made of the forms decode_step_replay can take (bf16-operand mode, skinny products
// for <= 64 utterances, the position row and the logistic fused): the conditions decode_step itself checks, asked up front
int32_t s2st_engine_decode_replay_supported(const s2st_engine* e) {
  if (!e || !e->dec_st.base || !e->dec_st.pe_alpha || !e->fast() || !e->PH || !e->use_skinny) return 0;
  const int B = e->dec_st.B, Cd = e->c.dec_dim;
  if (B < 1 || B > 64 || e->c.prenet_dim % 32 || e->c.out_dim % 32 || Cd % 64 || e->c.prenet_layers > 7) return 0;
  if (e->has_dec_ln && !e->use_ln_skinny) return 0;
  return 1;
}

int s2st_engine_decode_replay_begin(s2st_engine* e, const s2st_decode_replay* r, uint64_t seed0, void* stream) {
  if (!e || !r || !e->dec_st.base || !e->dec_st.pe_alpha) return S2ST_ERR_ARG;
  return s2st_decode_replay_init(r->step, r->seeds, r->cur_feat, (long)e->dec_st.B * e->c.out_dim, r->pe_cur, e->dec_st.pe_alpha,
                                 e->c.dec_dim, seed0, (hipStream_t)stream);
}

int s2st_engine_decode_step_replay(s2st_engine* e, const s2st_decode_replay* r, const int32_t* self_klen, float* workspace,
                                   int64_t workspace_floats, void* stream) {
  if (!e || !e->P || !r || !r->step || !r->seeds || !r->cur_feat || !r->cur_eos || !r->pe_cur) return S2ST_ERR_ARG;
  if (!e->fast() || !e->PH || !e->dec_st.pe_alpha) return S2ST_ERR_SHAPE;
  s2st_engine::Dec keep = e->dec_st;
  s2st_batch bt_keep = e->bt;
  e->reset_call();
  e->dec_st = keep;
  e->bt = bt_keep;
  e->dry = false;
  e->ws = workspace;
  e->ws_cap = workspace_floats;
  e->st_ = (hipStream_t)stream;
  e->skws = nullptr; e->skws_n = 0; e->skws_side = nullptr;
  e->replay_ = r;
  // (step 0 / seed 0 on the host side: every step-dependent value comes from *r inside the kernels)
  int rc = e->decode_step(0, r->cur_feat, nullptr, self_klen, 0, r->cur_feat, r->cur_eos, r->cur_attn);
  e->replay_ = nullptr;
  e->tape.clear();
  return rc;
}

int s2st_engine_decode_replay_commit(s2st_engine* e, const s2st_decode_replay* r, uint64_t seed0, float thr, int32_t max_iter,
                                     int32_t* finished, int32_t* out_lens, int32_t* klen_next, int32_t* n_done,
                                     float* feat_all, float* eos_all, float* attn_all, void* stream) {
  if (!e || !r || !e->dec_st.base || !e->dec_st.pe_alpha) return S2ST_ERR_ARG;
  return s2st_decode_replay_commit(r->step, r->seeds, r->cur_feat, r->cur_eos, r->cur_attn, r->pe_cur, e->dec_st.pe_alpha,
                                   e->dec_st.maxT + 2, e->c.dec_dim, seed0, thr, max_iter, e->dec_st.B, e->c.out_dim, e->dec_st.E,
                                   finished, out_lens, klen_next, n_done, feat_all, eos_all, attn_all, (hipStream_t)stream);
}

int s2st_engine_postnet_eval(s2st_engine* e, const float* feat, int32_t B, int32_t D, float* post_out,
                             float* workspace, int64_t workspace_floats, void* stream) {
  if (!e->P || !e->BUF || !feat || !post_out) return S2ST_ERR_ARG;
  s2st_engine::Dec keep = e->dec_st;
  e->reset_call();
  e->dec_st = keep;
  e->dry = false;
  e->ws = workspace;
  e->ws_cap = workspace_floats;
  e->st_ = (hipStream_t)stream;
  e->bt.training = 0;
  e->skws = nullptr; e->skws_n = 0; e->skws_side = nullptr;
  if (e->fast()) {
    if (!e->PH) return S2ST_ERR_ARG;
    int rc = s2st_cast_bf16_rows(e->P, e->n_params, e->PH, e->n_params, 1, (int)e->n_params, e->st_);
    if (rc) return rc;
  }
  std::vector<s2st_engine::ConvW> csp;
  for (auto& pc : e->post_conv) csp.push_back(e->make_conv_scratch(pc, false, false));
  Ten* f = e->newT(B * D, e->c.out_dim, const_cast<float*>(feat));
  f->needs_grad = false;
  e->postnet(f, B, D, false, csp, post_out);
  e->tape.clear();
  return e->err;
}

// ---- aux ASR / ST text decoder, forward only (beam search over an aux head: generate_for_s2st.py:107-111) -----
int s2st_engine_aux_decode(s2st_engine* e, int32_t which, const float* tap, const int32_t* enc_lens,
                           const int64_t* prev_tokens, const int32_t* positions, const int32_t* lens, const float* pe,
                           int32_t Bb, int32_t L, int32_t E, float* logits_out, float* workspace,
                           int64_t workspace_floats, void* stream) {
  if (!e || !e->P || !tap || !enc_lens || !prev_tokens || !positions || !lens || !pe || !logits_out) return S2ST_ERR_ARG;
  if ((which == 0 && !e->c.has_asr) || (which == 1 && !e->c.has_st) || which < 0 || which > 1) return S2ST_ERR_ARG;
  if (Bb <= 0 || L <= 0 || E <= 0) return S2ST_ERR_SHAPE;
  if (!workspace) return S2ST_ERR_WORKSPACE;
  e->reset_call();
  e->dry = false;
  e->ws = workspace;
  e->ws_cap = workspace_floats;
  e->st_ = (hipStream_t)stream;
  e->bt = s2st_batch{};
  e->bt.B = Bb; e->bt.E = E; e->bt.training = 0;
  e->bt.enc_lens = enc_lens;
  e->skws = nullptr; e->skws_n = 0; e->skws_side = nullptr;
  if (e->fast()) {
    if (!e->PH) return S2ST_ERR_ARG;
    int rc = s2st_cast_bf16_rows(e->P, e->n_params, e->PH, e->n_params, 1, (int)e->n_params, e->st_);
    if (rc) return rc;
  }
  hipStream_t keep_side = e->side_;
  e->side_ = nullptr;  // forward only, one stream
  Ten* t = e->newT(Bb * E, e->c.enc_dim, const_cast<float*>(tap));
  t->needs_grad = false;
  e->aux_decoder(which == 0 ? e->asr : e->st, t, (const long*)prev_tokens, positions, lens, Bb, L, pe, logits_out);
  e->side_ = keep_side;
  e->tape.clear();
  return e->err;
}

// ---- the same decoder step by step with key / value caches (include/s2st_hip.h) --------------------------------------------
namespace {
bool aux_inc_args_ok(const s2st_engine* e, int which) {
  return e && which >= 0 && which <= 1 && !((which == 0 && !e->c.has_asr) || (which == 1 && !e->c.has_st));
}
void aux_inc_enter(s2st_engine* e, float* workspace, int64_t workspace_floats, void* stream, int Bb, int E) {
  e->reset_call();
  e->dry = false;
  e->ws = workspace;
  e->ws_cap = workspace_floats;
  e->st_ = (hipStream_t)stream;
  e->bt = s2st_batch{};
  e->bt.B = Bb; e->bt.E = E; e->bt.training = 0;
  e->skws = nullptr; e->skws_n = 0; e->skws_side = nullptr;
}
}  // namespace

int64_t s2st_engine_aux_inc_state_floats(const s2st_engine* e, int32_t which, int32_t Bb, int32_t E, int32_t max_len) {
  if (!aux_inc_args_ok(e, which) || Bb <= 0 || E <= 0 || max_len <= 0) return S2ST_ERR_ARG;
  const AuxP& a = which == 0 ? e->asr : e->st;
  return 2 * s2st_engine::aux_inc_half(a, Bb, max_len) + (long)a.layers * Bb * E * 2 * a.d + 64;
}

int64_t s2st_engine_aux_inc_workspace(const s2st_engine* e, int32_t which, int32_t Bb, int32_t E) {
  if (!aux_inc_args_ok(e, which) || Bb <= 0 || E <= 0) return S2ST_ERR_ARG;
  const AuxP& a = which == 0 ? e->asr : e->st;
  // begin: bf16 copies of the tap and of every layer's K | V projection; a step: a few dozen [Bb][width] tensors
  long w = a.in_dim;
  for (long v : {(long)3 * a.d, (long)e->c.dec_ffn, (long)a.V, (long)a.out_dim, (long)e->c.enc_dim}) w = v > w ? v : w;
  return (long)Bb * E * (e->c.enc_dim + 2L * a.layers * a.d) + 96L * Bb * w + (1L << 20);
}

int s2st_engine_aux_inc_begin(s2st_engine* e, int32_t which, const float* tap, const int32_t* enc_lens, int32_t Bb, int32_t E,
                              int32_t max_len, float* state, float* workspace, int64_t workspace_floats, void* stream) {
  if (!aux_inc_args_ok(e, which) || !e->P || !tap || !enc_lens || !state) return S2ST_ERR_ARG;
  if (Bb <= 0 || E <= 0 || max_len <= 0) return S2ST_ERR_SHAPE;
  if (!workspace) return S2ST_ERR_WORKSPACE;
  aux_inc_enter(e, workspace, workspace_floats, stream, Bb, E);
  e->bt.enc_lens = enc_lens;
  if (e->fast()) {
    if (!e->PH) return S2ST_ERR_ARG;
    int rc = s2st_cast_bf16_rows(e->P, e->n_params, e->PH, e->n_params, 1, (int)e->n_params, e->st_);
    if (rc) return rc;
  }
  hipStream_t keep_side = e->side_;
  e->side_ = nullptr;  // forward only, one stream
  s2st_engine::AuxInc& S = e->aux_inc[which];
  S = s2st_engine::AuxInc{};
  S.base = state; S.Bb = Bb; S.E = E; S.maxT = max_len; S.enc_lens = enc_lens;
  Ten* t = e->newT(Bb * E, e->c.enc_dim, const_cast<float*>(tap));
  t->needs_grad = false;
  const int rc = e->aux_inc_begin(which == 0 ? e->asr : e->st, S, t);
  e->side_ = keep_side;
  e->tape.clear();
  return rc;
}

int s2st_engine_aux_inc_step(s2st_engine* e, int32_t which, int32_t step, const int64_t* tokens, const int32_t* reorder,
                             const int32_t* positions, const float* pe, float* logits_out, float* workspace,
                             int64_t workspace_floats, void* stream) {
  if (!aux_inc_args_ok(e, which) || !e->P || !tokens || !positions || !pe || !logits_out) return S2ST_ERR_ARG;
  if (!workspace) return S2ST_ERR_WORKSPACE;
  s2st_engine::AuxInc keep = e->aux_inc[which];
  if (!keep.base) return S2ST_ERR_ARG;
  s2st_engine::AuxInc keep_other = e->aux_inc[1 - which];
  aux_inc_enter(e, workspace, workspace_floats, stream, keep.Bb, keep.E);
  e->aux_inc[which] = keep;
  e->aux_inc[1 - which] = keep_other;
  e->bt.enc_lens = keep.enc_lens;
  if (e->fast() && !e->PH) return S2ST_ERR_ARG;  // (PH was refreshed by aux_inc_begin)
  hipStream_t keep_side = e->side_;
  e->side_ = nullptr;
  const int rc = e->aux_inc_step(which == 0 ? e->asr : e->st, e->aux_inc[which], step, (const long*)tokens, reorder, positions, pe,
                                 logits_out);
  e->side_ = keep_side;
  e->tape.clear();
  return rc;
}

int64_t s2st_engine_aux_decode_workspace(s2st_engine* e, int32_t which, int32_t Bb, int32_t L, int32_t E) {
  if (!e || which < 0 || which > 1 || (which == 0 && !e->c.has_asr) || (which == 1 && !e->c.has_st)) return S2ST_ERR_ARG;
  e->reset_call();
  e->dry = true;
  e->ws = reinterpret_cast<float*>(0x10000);
  e->ws_cap = (long)1 << 50;
  e->st_ = nullptr;
  e->bt = s2st_batch{};
  e->bt.B = Bb; e->bt.E = E; e->bt.training = 0;
  e->skws = nullptr; e->skws_n = 0; e->skws_side = nullptr;
  hipStream_t keep_side = e->side_;
  e->side_ = nullptr;
  Ten* t = e->newT(Bb * E, e->c.enc_dim, reinterpret_cast<float*>(0x10000));
  t->needs_grad = false;
  e->aux_decoder(which == 0 ? e->asr : e->st, t, nullptr, nullptr, nullptr, Bb, L, nullptr, reinterpret_cast<float*>(0x10000));
  e->side_ = keep_side;
  const long peak = e->ws_peak;
  const int err = e->err;
  e->reset_call();
  e->dry = false;
  return err ? (int64_t)err : (int64_t)peak + 1024;
}

// ---- HuBERT front end ---------------------------------------------------------------------------
int s2st_hubert_create(const s2st_hubert_config* cfg, s2st_engine** out) {
  if (!cfg || !out || cfg->n_conv < 1 || cfg->n_conv > 8) return S2ST_ERR_ARG;
  if (cfg->embed % cfg->heads || cfg->embed % cfg->conv_pos_groups || cfg->conv_dim[0] % 4 ||
      (cfg->embed / cfg->conv_pos_groups) % 4 || cfg->embed % 4)
    return S2ST_ERR_SHAPE;
  if (!cfg->precise) {
    for (int i = 0; i < cfg->n_conv; ++i)
      if (cfg->conv_dim[i] % 8) return S2ST_ERR_SHAPE;
    if (cfg->embed % 8 || cfg->ffn % 8 || (cfg->embed / cfg->conv_pos_groups) % 8) return S2ST_ERR_SHAPE;
  }
  s2st_engine* e = new s2st_engine();
  e->is_hubert = true;
  e->hc = *cfg;
  e->c = s2st_model_config{};
  e->c.precise = cfg->precise;
  e->c.enc_heads = cfg->heads;
  e->c.enc_dim = cfg->embed;
  e->ffn_act = 2;
  e->f32_operands = s2st_env_on("S2ST_F32_OPERANDS");
  e->use_flash = !(s2st_env_on("S2ST_NO_FLASH"));
  e->build_params_hubert();
  // (a process that replays HIP graphs can carry a stale "last error" of the runtime's own capture-time queries: the preload's
  //  launch checks must see their own errors only)
  (void)hipGetLastError();
  if (!cfg->precise && (s2st_gemm_bf16_preload(nullptr) != 0 || s2st_flash_attn_preload(nullptr) != 0)) { delete e; return S2ST_ERR_LAUNCH; }
  *out = e;
  return 0;
}

int32_t s2st_hubert_out_frames(const s2st_engine* e, int32_t n_samples) { return e->hubert_frames(n_samples); }

int64_t s2st_hubert_workspace_floats(s2st_engine* e, int32_t B, int32_t N) {
  if (!e->is_hubert) return S2ST_ERR_ARG;
  e->reset_call();
  e->dry = true;
  e->ws = reinterpret_cast<float*>(0x10000);
  e->ws_cap = (long)1 << 50;
  e->st_ = nullptr;
  int rc = e->forward_hubert(nullptr, nullptr, B, N, nullptr);
  long peak = e->ws_peak;
  e->reset_call();
  e->dry = false;
  return rc ? (int64_t)rc : (int64_t)peak + 1024;
}

int s2st_hubert_forward(s2st_engine* e, const float* wave, const int32_t* frame_lens, int32_t B, int32_t N, float* out,
                        float* workspace, int64_t workspace_floats, void* stream) {
  if (!e->is_hubert || !e->P || !wave || !frame_lens || !out) return S2ST_ERR_ARG;
  e->reset_call();
  e->dry = false;
  e->ws = workspace;
  e->ws_cap = workspace_floats;
  e->st_ = (hipStream_t)stream;
  const bool fm = e->fast();
  if (fm && !e->PH) return S2ST_ERR_ARG;
  // frozen weights: the bf16 copy is refreshed unless the caller vouches for it (s2st_engine_bf16_is_fresh before this call:
  // the host side tracks the parameter tensor's version -- 0.1 ms of the 6.9 ms forward)
  if (fm && !e->ph_fresh) {
    int rc = s2st_cast_bf16_rows(e->P, e->n_params, e->PH, e->n_params, 1, (int)e->n_params, e->st_);
    if (rc) return rc;
  }
  e->ph_fresh = false;
  int rc = e->forward_hubert(wave, frame_lens, B, N, out);
  e->tape.clear();  // forward only: the front end is frozen (s2st_transformer.py:245-249)
  return rc;
}

}  // extern "C"
