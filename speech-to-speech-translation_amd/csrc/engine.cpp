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

  // ------------------------------------------------------------------------------------
  long add(const std::string& name, std::vector<int> shape, int is_buffer = 0) {
    PInfo p;
    p.name = name;
    p.ndim = (int)shape.size();
    p.numel = 1;
    for (int i = 0; i < 4; ++i) p.shape[i] = i < p.ndim ? shape[i] : 1;
    for (int s : shape) p.numel *= s;
    long& top = is_buffer ? n_buffers : n_params;
    p.off = top;
    top += (p.numel + 7) / 8 * 8;  // every tensor 32-byte aligned (16 bytes in the bf16 copy)
    p.is_buffer = is_buffer;
    infos.push_back(p);
    return p.off;
  }
  LinP add_lin(const std::string& pre, int N, int K, bool bias = true) {
    LinP l;
    l.N = N; l.K = K;
    l.w = add(pre + ".weight", {N, K});
    l.b = bias ? add(pre + ".bias", {N}) : -1;
    reg_wt(l.w, N, K);
    return l;
  }
  LNP add_ln(const std::string& pre, int C) {
    LNP l;
    l.C = C;
    l.g = add(pre + ".weight", {C});
    l.b = add(pre + ".bias", {C});
    return l;
  }
  AttnP add_self_attn(const std::string& pre, int C) {
    AttnP a;
    a.kvq_w = add(pre + ".k_proj.weight", {C, C});
    add(pre + ".v_proj.weight", {C, C});
    add(pre + ".q_proj.weight", {C, C});
    a.kvq_b = add(pre + ".k_proj.bias", {C});
    add(pre + ".v_proj.bias", {C});
    add(pre + ".q_proj.bias", {C});
    a.out_w = add(pre + ".out_proj.weight", {C, C});
    a.out_b = add(pre + ".out_proj.bias", {C});
    reg_wt(a.kvq_w, 3 * C, C);
    reg_wt(a.out_w, C, C);
    return a;
  }
  XAttnP add_cross_attn(const std::string& pre, int C, int Cenc) {
    XAttnP a;
    a.kv_w = add(pre + ".k_proj.weight", {C, Cenc});
    add(pre + ".v_proj.weight", {C, Cenc});
    a.kv_b = add(pre + ".k_proj.bias", {C});
    add(pre + ".v_proj.bias", {C});
    a.q_w = add(pre + ".q_proj.weight", {C, C});
    a.q_b = add(pre + ".q_proj.bias", {C});
    a.out_w = add(pre + ".out_proj.weight", {C, C});
    a.out_b = add(pre + ".out_proj.bias", {C});
    reg_wt(a.kv_w, 2 * C, Cenc);
    reg_wt(a.q_w, C, C);
    reg_wt(a.out_w, C, C);
    return a;
  }
  DecLayerP add_dec_layer(const std::string& pre, int C, int ffn, int Cenc) {
    DecLayerP l;
    l.sa = add_self_attn(pre + ".self_attn", C);
    l.ln1 = add_ln(pre + ".self_attn_layer_norm", C);
    l.xa = add_cross_attn(pre + ".encoder_attn", C, Cenc);
    l.ln2 = add_ln(pre + ".encoder_attn_layer_norm", C);
    l.fc1 = add_lin(pre + ".fc1", ffn, C);
    l.fc2 = add_lin(pre + ".fc2", C, ffn);
    l.ln3 = add_ln(pre + ".final_layer_norm", C);
    return l;
  }
  // out_dim: the decoder's output width (DecoderConfig.output_dim: 512 for the aux heads whatever their width,
  // transformer_config.py:63-68; decoder_embed_dim for the s2t model's own decoder, s2t_transformer_me.py:527-529)
  AuxP add_aux(const std::string& pre, int V, int in_dim, int d, int layers, int out_dim = 512) {
    AuxP a;
    a.V = V; a.in_dim = in_dim; a.d = d; a.layers = layers; a.out_dim = out_dim;
    a.embed = add(pre + ".embed_tokens.weight", {V, in_dim});
    a.proj_in = d != in_dim ? add(pre + ".project_in_dim.weight", {d, in_dim}) : -1;
    for (int i = 0; i < layers; ++i)
      a.L.push_back(add_dec_layer(pre + ".layers." + std::to_string(i), d, c.dec_ffn, c.enc_dim));
    a.has_ln = c.dec_pre_ln != 0;
    if (a.has_ln) a.ln = add_ln(pre + ".layer_norm", d);
    a.proj_out = d != out_dim ? add(pre + ".project_out_dim.weight", {out_dim, d}) : -1;
    a.out_proj = add(pre + ".output_projection.weight", {V, out_dim});
    return a;
  }

  long enc_spk = -1, dec_spk = -1;  // speaker-embedding tables (n_speakers > 0)
  // frozen tables (Embedding.from_pretrained(freeze=True), tasks/s2s_translation.py:161-171) live in the BUFFER arena like
  // the BatchNorm statistics: the reference leaves them out of the optimizer, so neither Adam's sweep over the
  // parameter arena nor weight decay may touch them
  const float* spk_tab(long off) const { return (c.spk_frozen ? BUF : P) + off; }
  void touch_spk(long off_end) { if (!c.spk_frozen) touch(off_end); }
  LinP enc_spk_proj{-1, -1, 0, 0};  // t2s text encoder: spk_emb_proj over cat[x, emb] (t2s_transformer.py:43-46, 107-111)
  // t2s text encoder front
  long enc_embed = -1, enc_pos_alpha = -1;
  std::vector<ConvP> enc_conv;
  std::vector<BNP> enc_bn;
  LinP enc_prenet_proj;

  void build_params() {
    const int C = c.enc_dim, Cd = c.dec_dim;
    // forward-use order == arena order (see file header)
    if (c.text_input) {
      enc_embed = add("encoder.embed_tokens.weight", {c.src_vocab, C});
      for (int i = 0; i < c.enc_conv_layers; ++i) {
        std::string pre = "encoder.prenet." + std::to_string(i);
        enc_conv.push_back(ConvP{add(pre + ".0.weight", {C, C, c.enc_conv_k}), add(pre + ".0.bias", {C}), C, C, c.enc_conv_k});
        BNP bn;
        bn.C = C;
        bn.g = add(pre + ".1.weight", {C});
        bn.b = add(pre + ".1.bias", {C});
        bn.rm = add(pre + ".1.running_mean", {C}, 1);
        bn.rv = add(pre + ".1.running_var", {C}, 1);
        enc_bn.push_back(bn);
      }
      enc_prenet_proj = add_lin("encoder.prenet_proj", C, C);
      enc_pos_alpha = add("encoder.pos_emb_alpha", {1});
    } else {
    sub[0] = ConvP{add("encoder.subsample.conv_layers.0.weight", {c.conv_channels, c.in_dim, c.conv_k}),
                   add("encoder.subsample.conv_layers.0.bias", {c.conv_channels}), c.conv_channels,
                   c.in_dim, c.conv_k};
    sub[1] = ConvP{add("encoder.subsample.conv_layers.1.weight", {2 * C, c.conv_channels / 2, c.conv_k}),
                   add("encoder.subsample.conv_layers.1.bias", {2 * C}), 2 * C, c.conv_channels / 2,
                   c.conv_k};
    if (c.n_speakers > 0) enc_spk = add("encoder.embed_speaker.weight", {c.n_speakers, C}, c.spk_frozen ? 1 : 0);
    }
    for (int i = 0; i < c.enc_layers; ++i) {
      std::string pre = "encoder.transformer_layers." + std::to_string(i);
      EncLayerP l;
      l.sa = add_self_attn(pre + ".self_attn", C);
      l.ln1 = add_ln(pre + ".self_attn_layer_norm", C);
      l.fc1 = add_lin(pre + ".fc1", c.enc_ffn, C);
      l.fc2 = add_lin(pre + ".fc2", C, c.enc_ffn);
      l.ln2 = add_ln(pre + ".final_layer_norm", C);
      enc.push_back(l);
    }
    has_enc_ln = c.enc_pre_ln != 0;
    if (has_enc_ln) enc_ln = add_ln("encoder.layer_norm", C);
    if (c.text_input && c.n_speakers > 0) {
      // the table is spk_dim wide here (task.get_speaker_embeddings: Embedding(len(speaker_to_id), speaker_embed_dim))
      enc_spk = add("encoder.embed_speaker.weight", {c.n_speakers, c.spk_dim}, c.spk_frozen ? 1 : 0);
      enc_spk_proj = add_lin("encoder.spk_emb_proj", C, C + c.spk_dim);
    }
    if (c.s2t_mode) {
      // s2t_transformer_hubert: speech encoder + ONE full-width text decoder (s2t_transformer_me.py:266-283, 473-492)
      s2t = add_aux("decoder", c.tgt_vocab, Cd, Cd, c.dec_layers, Cd);
      return;
    }
    if (c.has_asr) asr_norm = add_ln("encoder.aux_asr_norm", C);
    if (c.has_st) st_norm = add_ln("encoder.aux_st_norm", C);
    pos_alpha = add("decoder.pos_emb_alpha", {1});
    if (c.n_speakers > 0 && !c.text_input)  // (the t2s decoder takes `speaker` and ignores it: t2s_transformer.py:172-176)
      dec_spk = add("decoder.embed_speaker.weight", {c.n_speakers, c.out_dim}, c.spk_frozen ? 1 : 0);
    for (int i = 0; i < c.prenet_layers; ++i)
      prenet.push_back(add_lin("decoder.prenet.0.layers." + std::to_string(i) + ".0",
                               c.prenet_dim, i == 0 ? c.out_dim : c.prenet_dim));
    prenet.push_back(add_lin("decoder.prenet.1", Cd, c.prenet_dim));
    for (int i = 0; i < c.dec_layers; ++i)
      dec.push_back(add_dec_layer("decoder.transformer_layers." + std::to_string(i), Cd, c.dec_ffn, C));
    has_dec_ln = c.dec_pre_ln != 0;
    if (has_dec_ln) dec_ln = add_ln("decoder.layer_norm", Cd);
    feat_proj = add_lin("decoder.feat_proj", c.out_dim, Cd);
    eos_proj = add_lin("decoder.eos_proj", 1, Cd);
    for (int i = 0; i < c.postnet_layers; ++i) {
      int ci = i == 0 ? c.out_dim : c.postnet_dim;
      int co = i == c.postnet_layers - 1 ? c.out_dim : c.postnet_dim;
      std::string pre = "decoder.postnet.convolutions." + std::to_string(i);
      ConvP cv{add(pre + ".0.weight", {co, ci, c.postnet_k}), add(pre + ".0.bias", {co}), co, ci, c.postnet_k};
      post_conv.push_back(cv);
      BNP bn;
      bn.C = co;
      bn.g = add(pre + ".1.weight", {co});
      bn.b = add(pre + ".1.bias", {co});
      bn.rm = add(pre + ".1.running_mean", {co}, 1);
      bn.rv = add(pre + ".1.running_var", {co}, 1);
      post_bn.push_back(bn);
    }
    // (t2s_transformer: the head reads the decoder's feature_out, t2s_transformer.py:168-170, 258)
    if (c.has_ctc) ctc_proj = add_lin("decoder.ctc_proj", c.src_vocab, c.text_input ? c.out_dim : C);
    if (c.has_ctc_tgt) ctc_proj_tgt = add_lin("decoder.ctc_proj_tgt", c.tgt_vocab, Cd);  // mtl variant
    // aux decoders: embedding dims follow the reference's in-place args mutation
    // (s2st_transformer.py:492-493, 541-542, 669-678)
    int cur = Cd;
    if (c.has_asr) {
      asr = add_aux("aux_asr_decoder", c.src_vocab, cur, c.asr_dim, c.asr_layers);
      cur = c.asr_dim;
    }
    if (c.has_st) st = add_aux("aux_st_decoder", c.tgt_vocab, cur, c.st_dim, c.st_layers);
  }

  // ------------------------------------------------------------------------------------
  // arena
  float* alloc(long n, bool zero = false) {
    n = (n + 63) / 64 * 64;
    float* p = nullptr;
    if (ws_top + n > ws_cap) {
      if (!dry) { oom = true; err = S2ST_ERR_WORKSPACE; }
    } else {
      p = ws + ws_top;
    }
    ws_top += n;
    if (ws_top > ws_peak) ws_peak = ws_top;
    if (p && zero && !dry) hipMemsetAsync(p, 0, sizeof(float) * n, st_);
    return p;
  }
  Ten* newT(int rows, int cols, float* ext = nullptr, bool f32 = true) {
    Ten* t = new Ten();
    t->rows = rows; t->cols = cols;
    t->d = ext ? ext : (f32 ? alloc(t->n()) : nullptr);
    tens.push_back(t);
    return t;
  }
  // gradient buffer of t: first request allocates it (acc = false: caller must overwrite),
  // later requests accumulate
  float* gradbuf(Ten* t, bool& acc) {
    if (t->g) { acc = true; return t->g; }
    t->g = alloc(t->n());
    acc = false;
    return t->g;
  }
  bf16raw* alloc_h(long n) { return reinterpret_cast<bf16raw*>(alloc((n + 1) / 2)); }
  bool live() const { return !dry && !oom && err == 0; }
  // bf16 copy of an activation / of its gradient (cast on first use unless the producer made it)
  bf16raw* half_of(Ten* t) {
    if (!t->h) {
      t->h = alloc_h((long)t->rows * t->hld());
      if (live()) { sync_chains(); chk(s2st_cast_bf16_rows(t->d, t->cols, t->h, t->hld(), t->rows, t->cols, st_)); }
    }
    return t->h;
  }
  bf16raw* ghalf_of(Ten* t) {
    if (!t->gh) {
      t->gh = alloc_h((long)t->rows * t->hld());
      if (live()) { sync_chains(); chk(s2st_cast_bf16_rows(t->g, t->cols, t->gh, t->hld(), t->rows, t->cols, st_)); }
    }
    return t->gh;
  }
  bf16raw* cast_buf(const float* x, long n) {  // whole-buffer twin (halo images, conv weights)
    bf16raw* y = alloc_h((n + 7) / 8 * 8);
    if (live()) chk(s2st_cast_bf16_rows(x, n, y, (n + 7) / 8 * 8, 1, (int)n, st_));
    return y;
  }
  // bf16 twin of a halo image [B][T + 2 pad][C] with zero halos; fast mode never reads the fp32 image's halos, so the
  // callers do not clear it (alloc(n, !fast()))
  // (plain: img is the plain rows [B * T][C] -- the first convolution of a stack needs no fp32 image at all)
  bf16raw* cast_halo(const float* img, int B, int T, int pad, int C, bool plain = false) {
    bf16raw* y = alloc_h(((long)B * (T + 2 * pad) * C + 7) / 8 * 8);
    if (live()) chk(s2st_cast_bf16_halo(img, y, B, T, pad, C, st_, plain ? 1 : 0));
    return y;
  }
  void chk(int rc) { if (rc && !err) err = rc; }
  // Dropout sites: the seed of the n-th site of a forward is a function of (batch seed, n).  s2st_engine_site_log(e, 1)
  // makes the forward also RECORD every site -- seed, kind, p, the element geometry its mask is indexed by and where in
  // the model it sits -- so that a test can regenerate the keep masks (s2st_dropout_f32 over ones) and hand them to the CPU
  // oracle: parity with the recipe's dropouts ON (tests/test_dropout_parity.py).  Nothing on the data path reads the log.
  bool site_log_on = false;
  std::vector<s2st_dropout_site> site_log;
  char site_ctx[24] = "";
  void set_ctx(const char* fmt, int i = 0) { snprintf(site_ctx, sizeof site_ctx, fmt, i); }
  uint64_t next_seed(int kind, float p, long d0, long d1 = 0, long d2 = 0, long d3 = 0, long d4 = 0) {
    const uint64_t s = seed * 0x100000001B3ULL + (++site) * 0x9E3779B97F4A7C15ULL;
    if (site_log_on) {
      s2st_dropout_site r{};
      r.seed = s; r.kind = kind; r.p = p;
      r.dims[0] = d0; r.dims[1] = d1; r.dims[2] = d2; r.dims[3] = d3; r.dims[4] = d4;
      snprintf(r.ctx, sizeof r.ctx, "%s", site_ctx);
      int ord = 0;
      for (const s2st_dropout_site& q : site_log) ord += (q.kind == kind && !strcmp(q.ctx, r.ctx)) ? 1 : 0;
      r.ordinal = ord;
      site_log.push_back(r);
    }
    return s;
  }
  void mark() { marks.push_back(Mark{tape.size(), param_watermark}); }
  void touch(long off_end) {
    if (off_end > param_watermark) param_watermark = off_end;
    if (adam_pending && st_ != side_) adam_wait_upto(off_end);
  }

  // ---- optimizer update overlapped with the next forward (s2st_engine_adam_overlapped) ------------------------------
  // The fused scale / clip / Adam kernel runs in chunks of the arena on the SECOND stream, one event per chunk.  The
  // arena is laid out in forward-use order and every op announces the parameters it is about to read (touch()), so the
  // next forward on the data-path stream waits for exactly the chunks it needs, when it needs them; work the forward
  // puts on the second stream (weight transposes, post-net weight layouts, hoisted K|V projections, aux heads) is
  // ordered behind the update by the stream itself.
  std::vector<hipEvent_t> adam_ev;
  std::vector<long> adam_lo;      // first element of chunk i
  int adam_next = 0;              // chunks [0, adam_next) have been waited for by the data-path stream
  bool adam_pending = false;
  void adam_wait_upto(long off_end) {
    if (!live()) return;
    while (adam_next < (int)adam_lo.size() && adam_lo[adam_next] < off_end) {
      hipStreamWaitEvent(st_, adam_ev[adam_next], 0);  // (st_ is the data-path stream here: touch() skips the second one)
      chains_wait(adam_ev[adam_next]);  // (a second chain forked earlier reads the same parameters)
      ++adam_next;
    }
    if (adam_next >= (int)adam_lo.size()) adam_pending = false;
  }
  int adam_overlapped(float* m, float* v, const float* sumsq, int nparts, float gmul, const float* gmul_dev, float max_norm,
                      float lr, float b1, float b2, float eps, float wd, int step, float* gnorm_out, int* skipped, int use_ph,
                      int nchunks, hipStream_t main) {
    if (!P || !G || n_params <= 0) return S2ST_ERR_ARG;
    if (nchunks < 1) nchunks = 1;
    if (nchunks > 64) nchunks = 64;
    ensure_side();
    hipStream_t saved = st_;
    st_ = main;
    hipStream_t a = side_ ? fork_side() : main;  // behind the norm's partial sums (and everything else) on `main`
    st_ = saved;
    while ((int)adam_ev.size() < nchunks) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return S2ST_ERR_LAUNCH;
      adam_ev.push_back(e);
    }
    adam_lo.assign(nchunks, 0);
    int rc = 0;
    for (int c2 = 0; c2 < nchunks && !rc; ++c2) {
      const long lo = (n_params * c2 / nchunks) / 64 * 64, hi = c2 + 1 == nchunks ? n_params : (n_params * (c2 + 1) / nchunks) / 64 * 64;
      adam_lo[c2] = lo;
      // (every chunk folds the norm's partials itself; only the first one reports the norm and counts a skipped update)
      rc = s2st_adam(P + lo, G + lo, m + lo, v + lo, hi - lo, sumsq, gmul, gmul_dev, max_norm, lr, b1, b2, eps, wd, step,
                     c2 == 0 ? gnorm_out : nullptr, a, (use_ph && PH) ? reinterpret_cast<uint16_t*>(PH) + lo : nullptr,
                     c2 == 0 ? skipped : nullptr, nparts, 1);
      if (side_) hipEventRecord(adam_ev[c2], a);
    }
    adam_next = 0;
    adam_pending = side_ != nullptr && rc == 0;
    return rc;
  }
  // `stream` waits for the whole update (callers that read parameters outside the engine)
  void adam_wait_all(hipStream_t stream) {
    if (!adam_pending) return;
    for (int i = adam_next; i < (int)adam_lo.size(); ++i) hipStreamWaitEvent(stream, adam_ev[i], 0);
    adam_next = (int)adam_lo.size();
    adam_pending = false;
  }

  // ------------------------------------------------------------------------------------
  // weight-gradient GEMMs waiting for their group launch (S2ST_NO_WGRAD_GROUP=1: A/B switch, one launch each)
  bool group_wgrad = true;
  int group_flush_at = 6;  // S2ST_WGRAD_GROUP=<n>: problems per launch (<= S2ST_GROUP_MAX); measured 2 .. 8 on the bench
                           // workload: 12.25 / 10.99 / 10.56 / 10.44 / 10.46 ms per step for 2 / 3 / 4 / 6 / 8
  std::vector<GemmArgs> pending_wgrad;
  void push_wgrad(const GemmArgs& g) {
    for (const GemmArgs& p : pending_wgrad)
      if (p.C.p == g.C.p) { flush_wgrad(); break; }  // two sums into one matrix must not share a launch
    pending_wgrad.push_back(g);
    if ((int)pending_wgrad.size() >= group_flush_at) flush_wgrad();
  }
  void flush_wgrad() {
    if (pending_wgrad.empty()) return;
    if (live()) {
      // everything the products read was enqueued on st_ before this point
      if (!side_) sync_chains();
      hipStream_t s = (side_ && st_ != side_) ? fork_side() : st_;
      // S2ST_TIMING_SKIP_WGRAD=1 (-DS2ST_EXPERIMENTAL builds only: it makes the gradients WRONG): a timing experiment --
      // how much of the step is the weight-gradient products' share of the chip
#ifdef S2ST_EXPERIMENTAL
      static const bool skip = [] {
        const bool on = s2st_env_on("S2ST_TIMING_SKIP_WGRAD");
        if (on) fprintf(stderr, "[s2st] S2ST_TIMING_SKIP_WGRAD=1: weight-gradient products are SKIPPED -- gradients are WRONG, "
                                "timing experiments only\n");
        return on;
      }();
#else
      constexpr bool skip = false;
#endif
      if (!skip) chk(s2st_gemm_bf16_group(pending_wgrad.data(), (int)pending_wgrad.size(), s));
    }
    pending_wgrad.clear();
  }

  // layer-norm parameter gradients: the backward row kernels leave column-sum partials; one batched fold per backward
  // segment (or per S2ST_LNFOLD_MAX layer norms) adds them to the gradient arena.  Parameter gradients only: the fold
  // runs on the second stream behind everything enqueued on the stream the row kernels ran on.
  s2st_lnfold_table pending_lnfold{};
  bool ln_bwd_split = false;  // S2ST_LN_BWD_SPLIT=1 (A/B switch): round 2's separate parameter-gradient pass
  void flush_lnfold() {
    if (pending_lnfold.n == 0) return;
    if (live()) {
      if (!side_) sync_chains();
      hipStream_t s = (side_ && st_ != side_) ? fork_side() : st_;
      chk(s2st_layernorm_bwd_fold(pending_lnfold, s));
    }
    pending_lnfold = s2st_lnfold_table{};
  }
  // out[c] += sum_b part[b][c] (b in order) joins the segment's batched fold
  void add_fold(const float* part, int nblocks, int cols, float* out) {
    if (!part || nblocks <= 0 || cols <= 0) return;
    if (pending_lnfold.n == S2ST_LNFOLD_MAX) flush_lnfold();
    chk(s2st_fold_add(pending_lnfold, part, nblocks, cols, 1, out, nullptr, nullptr));
  }

  // ------------------------------------------------------------------------------------
  // op: y = [resid +] dropout(act(x W^T + b))
  // only_h: the caller guarantees every consumer reads the bf16 copy (fast mode): no fp32 result is
  // allocated or written
  // resid_row (AR decoding only, skinny path): ONE row added to every output row (the step's alpha-scaled position)
  Ten* linear(Ten* x, long w, long b, int N, int K, int act = 0, float drop_p = 0.f,
              Ten* resid = nullptr, float* ext_out = nullptr, bool only_h = false, const float* resid_row = nullptr) {
    const int M = x->rows;
    const bool fm = fast();
    // AR decoding: a handful of rows (one per utterance) -- the skinny kernel converts x in registers, so neither
    // a bf16 copy of the input nor one of the output is made (fp32 in, fp32 out)
    const bool skinny = fm && !bt.training && use_skinny && M <= S2ST_SKINNY_MAX_ROWS && K % 32 == 0 && x->d && x->cols == K &&
                        (act == 0 || act == 1 || act == 3);
    only_h = only_h && fast() && use_only_h && N % 8 == 0 && !ext_out && !resid && !skinny;
    Ten* y = newT(M, N, ext_out, !only_h);
    touch(w + (long)N * K);
    if (b >= 0) touch(b + N);
    const uint64_t sd = drop_p > 0.f ? next_seed(S2ST_SITE_LINEAR, drop_p, M, N) : 0;
    if (skinny) {
      const uint64_t* seed_ptr = nullptr;
      if (replay_ && drop_p > 0.f) {
        if (site < 1 || site > 8) { if (!err) err = S2ST_ERR_SHAPE; return y; }
        seed_ptr = replay_->seeds + (site - 1);
      }
      if (live())
        chk(s2st_gemm_skinny(x->d, K, PH + w, K, y->d, N, b >= 0 ? P + b : nullptr, act, drop_p, sd,
                             resid ? resid->d : resid_row, resid ? N : 0, M, N, K, st_, nullptr, nullptr, 1e-5f, seed_ptr,
                             drop_p > 0.f ? dec_row_map : nullptr));
      return y;  // inference only: no tape entry
    }
    if (replay_ && drop_p > 0.f && !err) err = S2ST_ERR_SHAPE;  // (a dropout site off the skinny path: not replayable)
    const bf16raw* xh = fm ? half_of(x) : nullptr;
    // (a residual-stream output is read in fp32 by the next layer norm / residual add: no bf16 copy; a consumer that
    // does want one gets it from half_of())
    if (fm && N % 8 == 0 && !resid) y->h = alloc_h(y->n());  // (residual-stream outputs are only ever read as fp32)
    if (fm && act == 1 && y->h && !resid) { y->act_mode = 1; y->act_p = drop_p; y->act_bias = b; }
    if (fm && act == 0 && drop_p == 0.f && !resid && y->h) { y->lin_plain = true; y->act_bias = b; }
    if (fm && act == 0 && drop_p > 0.f && N % 8 == 0 && use_ln_fuse) {
      y->drop2_ok = true; y->drop2_p = drop_p; y->drop2_seed = sd; y->drop2_bias = b;
    }
    Part pt[2];
    const int np = chain_parts(M, pt);  // (two utterance-half chains: rows [r0, r0 + nr) on each chain's stream)
    if (live()) {
      for (int ci = 0; ci < np; ++ci) {
        const long r0 = pt[ci].r0;
        GemmArgs g{};
        g.A = fm ? gemm_rowmajor(xh + r0 * x->hld(), x->hld()) : gemm_rowmajor(x->d + r0 * x->cols, x->cols);
        g.B = fm ? gemm_rowmajor(PH + w, K) : gemm_rowmajor(P + w, K);
        g.C = gemm_out(y->d ? y->d + r0 * N : nullptr, N);
        g.C.h = y->h ? y->h + r0 * N : nullptr;
        g.ep = gemm_epi_default();
        g.ep.bias = b >= 0 ? P + b : nullptr;
        g.ep.act = act;
        g.ep.drop_p = drop_p;
        g.ep.seed = sd ^ pt[ci].salt;
        g.ep.resid = resid ? resid->d + r0 * N : nullptr;
        g.M = pt[ci].nr; g.N = N; g.K = K; g.batch = 1; g.zdiv = 1; g.precise = c.precise;
        chk(s2st_gemm(g, pt[ci].st));
      }
    }
    const bool region = in_region_;
    tape.push_back([=]() {
      const bool region_was = in_region_;
      in_region_ = region;
      struct Restore { bool& r; bool v; ~Restore() { r = v; } } restore_{in_region_, region_was};
      Part bp[2];
      const int nb = chain_parts(M, bp);
      if (!y->g && !y->gpre_h) return;  // nothing flowed back
      float* dy = y->g;
      if (resid && resid->needs_grad) {
        if (!resid->g) resid->g = dy;  // alias: every reader of dy runs before resid's producers
        else if (live())
          for (int ci = 0; ci < nb; ++ci)
            chk(s2st_axpy(dy + (long)bp[ci].r0 * N, resid->g + (long)bp[ci].r0 * N, (long)bp[ci].nr * N, 1.f, bp[ci].st));
      }
      float* dpre = dy;
      const int ldp = (N + 7) & ~7;
      const bf16raw* dph = nullptr;
      bool bias_done = false;
      if (y->gpre_h) {  // the consumer's data-gradient GEMM already applied f' and the bias gradient
        dph = y->gpre_h;
        bias_done = true;
      } else if (fm && N % 4 == 0) {
        // one pass: bf16 GEMM operand of f(dy) + bias gradient (no fp32 dpre is materialised)
        bf16raw* t = alloc_h((long)M * ldp);
        const int mode = act == 1 ? 1 : (drop_p > 0.f ? 2 : 0);
        // (bias sums in a fixed order: a bias whose gradient is mathematically zero -- key projections -- gets pure rounding
        // noise, which must repeat from run to run)
        float* part[2] = {nullptr, nullptr};
        for (int ci = 0; ci < nb; ++ci)
          part[ci] = (b >= 0 && ordered_sums) ? alloc(s2st_linear_bwd_prep_scratch_floats(M, N, ldp)) : nullptr;
        if (live()) {
          for (int ci = 0; ci < nb; ++ci) {
            const long r0 = bp[ci].r0;
            int slabs = 0;
            chk(s2st_linear_bwd_prep(dy + r0 * N, y->d ? y->d + r0 * N : nullptr, y->d ? nullptr : y->h + r0 * N, mode, drop_p,
                                     sd ^ bp[ci].salt, t + r0 * ldp, ldp, nullptr, b >= 0 ? G + b : nullptr, bp[ci].nr, N, bp[ci].st,
                                     part[ci], part[ci] ? &slabs : nullptr));
            if (part[ci]) add_fold(part[ci], slabs, N, G + b);
          }
        }
        dph = t;
        bias_done = true;
      } else {
        if (live()) sync_chains();  // (whole-tensor passes of the precise / odd-width path)
        if (act == 1) {
          dpre = alloc(y->n());
          if (live()) chk(s2st_relu_drop_bwd(dy, y->d, dpre, y->n(), drop_p, st_));
        } else if (drop_p > 0.f) {
          dpre = alloc(y->n());
          if (live()) chk(s2st_dropout(dy, dpre, y->n(), 1.f, drop_p, sd, 0, st_));
        }
        if (fm) {
          bf16raw* t = alloc_h((long)M * ldp);
          if (live()) chk(s2st_cast_bf16_rows(dpre, N, t, ldp, M, N, st_));
          dph = t;
        }
      }
      // (fixed-order bias sums, see above)
      float* cpart = (b >= 0 && !bias_done && ordered_sums) ? alloc(s2st_colsum_scratch_floats(M, N)) : nullptr;
      if (live()) {
        GemmArgs g{};  // dW[N][K] += dpre^T x
        g.A = fm ? gemm_colmajor(dph, ldp) : gemm_colmajor(dpre, N);
        g.B = fm ? gemm_colmajor(xh, x->hld()) : gemm_colmajor(x->d, x->cols);
        g.C = gemm_out(G + w, K);
        g.ep = gemm_epi_default();
        g.ep.accumulate = 1;
        // weight gradients go to the second stream
        constexpr bool on_main = false;
        g.M = N; g.N = K; g.K = M; g.batch = 1; g.zdiv = 1; g.precise = c.precise;
        if (fm && !on_main && group_wgrad && s2st_gemm_group_ok(g)) {
          // a layer's weight-gradient products leave together, as ONE persistent launch with K = tokens unsplit
          // (no slabs, no combine kernels): see flush_wgrad()
          push_wgrad(g);
        } else {
          if (!(fm && !on_main)) sync_chains();  // (a whole-batch product on the data-path stream)
          hipStream_t ws_st = fm && !on_main ? fork_side() : st_;
          g.ws = ws_for(ws_st); g.ws_floats = skws_n;
          chk(s2st_gemm(g, ws_st));
        }
        if (b >= 0 && !bias_done) {
          int slabs = 0;
          chk(s2st_colsum(dpre, N, M, N, G + b, 1, st_, cpart, cpart ? &slabs : nullptr));
          if (cpart) add_fold(cpart, slabs, N, G + b);
        }
      }
      if (x->needs_grad) {
        bool acc;
        float* dx = gradbuf(x, acc);
        if (fm && !acc && x->want_gh && x->hld() == x->cols) x->gh = alloc_h(x->n());
        const bool fuse_act = fm && !acc && x->act_mode == 1 && x->h && x->hld() == x->cols && use_act_fuse;
        if (fuse_act) x->gpre_h = alloc_h(x->n());
        float* cs_part[2] = {nullptr, nullptr};
        for (int ci = 0; ci < nb; ++ci)
          cs_part[ci] = (fuse_act && ordered_sums && x->act_bias >= 0) ? alloc((long)2 * ((M + 63) / 64) * K) : nullptr;
        if (live()) {
          if (nb == 2) ensure_forked();  // (a whole-batch pass above may have joined the chains)
          for (int ci = 0; ci < nb; ++ci) {
            const long r0 = bp[ci].r0;
            GemmArgs g{};  // dx[M][K] (+)= dpre W
            g.A = fm ? gemm_rowmajor(dph + r0 * ldp, ldp) : gemm_rowmajor(dpre + r0 * N, N);
            g.B = fm ? (has_wt(w, N, K) ? gemm_rowmajor(PHT + w, N) : gemm_colmajor(PH + w, K)) : gemm_colmajor(P + w, K);
            g.C = gemm_out(dx + r0 * x->cols, x->cols);
            if (fm && !acc && x->gh) g.C.h = x->gh + r0 * x->cols;  // the consumer (attention backward) reads dO as a GEMM operand
            g.ep = gemm_epi_default();
            if (fuse_act) {  // dx is the gradient w.r.t. a ReLU+dropout output: emit its pre-activation gradient
              g.C.p = nullptr;
              g.C.h = x->gpre_h + r0 * x->cols;
              g.ep.mask_y = x->h + r0 * x->cols;
              g.ep.mask_scale = x->act_p > 0.f ? 1.f / (1.f - x->act_p) : 1.f;
              g.ep.colsum = x->act_bias >= 0 ? G + x->act_bias : nullptr;
            }
            g.ep.accumulate = acc ? 1 : 0;
            g.ws = ws_for(bp[ci].st); g.ws_floats = skws_n;
            g.M = bp[ci].nr; g.N = K; g.K = N; g.batch = 1; g.zdiv = 1; g.precise = c.precise;
            if (fuse_act && g.ep.colsum && ordered_sums) {
              // the bias gradient of the masked layer as per-(row tile, wave row) partial rows (worst case: 64-row tiles)
              g.ep.colsum_part = cs_part[ci];
              int tile = 0;
              chk(s2st_gemm(g, bp[ci].st, &tile));
              const int bm = tile / 1000;
              if (bm > 0) add_fold(cs_part[ci], 2 * ((bp[ci].nr + bm - 1) / bm), K, g.ep.colsum);
              else if (!err) err = S2ST_ERR_LAUNCH;
            } else
            chk(s2st_gemm(g, bp[ci].st));
          }
        }
      }
    });
    set_aware();
    return y;
  }

  // inference with a handful of rows (AR decoding): y = act(LayerNorm(x) W^T + b) in ONE skinny launch (the
  // normalisation is applied while the rows are converted to bf16); otherwise layernorm() + linear()
  Ten* ln_linear(Ten* x, const LNP& ln, long w, long b, int N, int K, int act = 0, float* ext_out = nullptr) {
    const bool fused = fast() && !bt.training && use_skinny && use_ln_skinny && x->rows <= S2ST_SKINNY_MAX_ROWS && K % 64 == 0 &&
                       x->d && x->cols == K && (act == 0 || act == 1 || act == 3);
    if (!fused) return linear(layernorm(x, ln), w, b, N, K, act, 0.f, nullptr, ext_out);
    Ten* y = newT(x->rows, N, ext_out);
    touch(w + (long)N * K);
    if (b >= 0) touch(b + N);
    touch(ln.b + ln.C);
    if (live())
      chk(s2st_gemm_skinny(x->d, K, PH + w, K, y->d, N, b >= 0 ? P + b : nullptr, act, 0.f, 0, nullptr, 0, x->rows, N, K,
                           st_, P + ln.g, P + ln.b, 1e-5f));
    return y;
  }

  Ten* layernorm(Ten* x, const LNP& p, float* ext_out = nullptr, bool only_h = false) {
    only_h = only_h && fast() && use_only_h && x->cols % 8 == 0 && !ext_out;
    Ten* y = newT(x->rows, x->cols, ext_out, !only_h);
    float* mean = alloc(x->rows);
    float* rstd = alloc(x->rows);
    touch(p.b + p.C);
    if (fast() && x->cols % 8 == 0) y->h = alloc_h(y->n());
    Part pt[2];
    const int np = chain_parts(x->rows, pt);
    if (live())
      for (int ci = 0; ci < np; ++ci) {
        const long r0 = pt[ci].r0, o = r0 * x->cols;
        chk(s2st_layernorm_fwd(x->d + o, P + p.g, P + p.b, y->d ? y->d + o : nullptr, mean + r0, rstd + r0, pt[ci].nr, x->cols, 1e-5f,
                               pt[ci].st, y->h ? y->h + o : nullptr));
      }
    LNP pp = p;
    const bool region = in_region_;
    // first layer norm applied to x (forward order): its backward is the last contribution to x's gradient
    const bool fuse_cand = fast() && x->drop2_ok && !x->ln_seen && x->needs_grad;
    x->ln_seen = true;
    tape.push_back([=]() {
      if (!y->g) return;
      const bool region_was = in_region_;
      in_region_ = region;
      struct Restore { bool& r; bool v; ~Restore() { r = v; } } restore_{in_region_, region_was};
      Part bp[2];
      const int nb = chain_parts(x->rows, bp);
      bool acc;
      float* dx = gradbuf(x, acc);
      const bool fuse = fuse_cand && !x->gpre_h;
      float* scratch = alloc((long)s2st_layernorm_bwd_blocks(x->rows, x->cols) * (fuse ? 3 : 2) * x->cols);
      float* scratch1 = nb == 2 ? alloc((long)s2st_layernorm_bwd_blocks(x->rows, x->cols) * (fuse ? 3 : 2) * x->cols) : nullptr;
      bf16raw* dph = nullptr;
      if (fuse) dph = x->gpre_h = alloc_h(x->n());
      float* dbias = fuse && x->drop2_bias >= 0 ? G + x->drop2_bias : nullptr;
      if (live()) {
        if (nb == 2) ensure_forked();
        if (!ln_bwd_split) {
          // one row kernel on the data path (dx, the fused bf16 operand, and the column-sum partials of dgamma / dbeta /
          // dbias); the partials of the segment's layer norms are folded together (flush_lnfold)
          for (int ci = 0; ci < nb; ++ci) {
            const long r0 = bp[ci].r0, o = r0 * x->cols;
            float* sc = ci == 0 ? scratch : scratch1;
            chk(s2st_layernorm_bwd(y->g + o, x->d + o, P + pp.g, mean + r0, rstd + r0, dx + o, acc ? 1 : 0, G + pp.g, G + pp.b, sc,
                                   bp[ci].nr, x->cols, bp[ci].st, 3, dph ? dph + o : nullptr, x->drop2_p, x->drop2_seed ^ bp[ci].salt,
                                   dbias));
            if (pending_lnfold.n == S2ST_LNFOLD_MAX) flush_lnfold();
            chk(s2st_lnfold_add(pending_lnfold, sc, bp[ci].nr, x->cols, fuse ? 3 : 2, G + pp.g, G + pp.b, dbias));
          }
        } else {
          // S2ST_LN_BWD_SPLIT=1 (A/B switch): dx row kernel on the data path (per chain), then a second pass over dy and x
          // for the parameter gradients + its fold on the second stream (or behind it without one)
          auto pass = [&](int ci, int phase, hipStream_t st) {
            const long r0 = bp[ci].r0, o = r0 * x->cols;
            chk(s2st_layernorm_bwd(y->g + o, x->d + o, P + pp.g, mean + r0, rstd + r0, dx + o, acc ? 1 : 0, G + pp.g, G + pp.b,
                                   ci == 0 ? scratch : scratch1, bp[ci].nr, x->cols, st, phase, dph ? dph + o : nullptr, x->drop2_p,
                                   x->drop2_seed ^ bp[ci].salt, dbias));
          };
          for (int ci = 0; ci < nb; ++ci) pass(ci, 1, bp[ci].st);
          hipStream_t rs = side_ ? fork_side() : (sync_chains(), st_);
          for (int ci = 0; ci < nb; ++ci) pass(ci, 2, rs);
        }
      }
    });
    set_aware();
    return y;
  }

  // attention core.  q: [B*T] rows at qp (+ h*dh), ld ldq ; k/v rows [B*S] ; out [B*T][C]
  struct AttnIO {
    Ten* qt; int qoff, ldq;     // tensor holding q, column offset, row stride
    Ten* kt; int koff, ldk;
    Ten* vt; int voff, ldv;
  };
  Ten* attention(const AttnIO& io, int B, int T, int S, int H, int dh, const int* klen, int causal,
                 float drop_p, float* attn_mean_out /* [B][S][T] or null */) {
    const int C = H * dh;
    const int ld = (S + 7) / 8 * 8;
    const bool fm = fast();
    Ten* o = newT(B * T, C);
    o->want_gh = true;
    if (fm && C % 8 == 0) o->h = alloc_h(o->n());
    // fused path (attention.hip): no [B,H,T,S] tensors in HBM.  The head-averaged attention map of
    // the last decoder layer still needs the probabilities, so that one call stays unfused.
    if (fm && use_flash && s2st_flash_attn_supported(dh) && !attn_mean_out && o->h && io.ldq % 8 == 0 &&
        io.ldk % 8 == 0 && io.ldv % 8 == 0) {
      const uint64_t sd = drop_p > 0.f ? next_seed(S2ST_SITE_ATTN, drop_p, B, H, T, S, ld) : 0;
      float* lse = alloc((long)B * H * T);
      s2st_attn_args fa{};
      fa.q = half_of(io.qt) + io.qoff; fa.k = half_of(io.kt) + io.koff; fa.v = half_of(io.vt) + io.voff;
      fa.ldq = io.ldq; fa.ldk = io.ldk; fa.ldv = io.ldv;
      fa.o = o->d; fa.oh = o->h; fa.lse = lse; fa.klen = klen;
      fa.B = B; fa.H = H; fa.T = T; fa.S = S; fa.dh = dh; fa.causal = causal;
      fa.scale = 1.0f / sqrtf((float)dh); fa.drop_p = drop_p; fa.seed = sd; fa.ld_drop = ld;
      // one chain's share of the batch: utterances [b0, b0 + nbat) of every per-utterance array
      auto chain_args = [=](s2st_attn_args a, int b0, int nbat, uint64_t salt) {
        a.q += (long)b0 * T * a.ldq; a.k += (long)b0 * S * a.ldk; a.v += (long)b0 * S * a.ldv;
        a.o += (long)b0 * T * C; if (a.oh) a.oh += (long)b0 * T * C;
        a.lse += (long)b0 * H * T; if (a.klen) a.klen += b0;
        if (a.doh) a.doh += (long)b0 * T * C;
        if (a.dq) a.dq += (long)b0 * T * a.ldq; if (a.dk) a.dk += (long)b0 * S * a.ldk; if (a.dv) a.dv += (long)b0 * S * a.ldv;
        if (a.dqh) a.dqh += (long)b0 * T * a.ldq; if (a.dkh) a.dkh += (long)b0 * S * a.ldk; if (a.dvh) a.dvh += (long)b0 * S * a.ldv;
        a.B = nbat; a.seed ^= salt;
        return a;
      };
      Part pt[2];
      const int np = chain_parts(B * T, pt);
      if (live())
        for (int ci = 0; ci < np; ++ci) {
          const s2st_attn_args a = chain_args(fa, pt[ci].r0 / T, pt[ci].nr / T, pt[ci].salt);
          chk(s2st_flash_attn_fwd(&a, pt[ci].st));
        }
      AttnIO io3 = io;
      const bool region = in_region_;
      tape.push_back([=]() {
        if (!o->g) return;
        const bool region_was = in_region_;
        in_region_ = region;
        struct Restore { bool& r; bool v; ~Restore() { r = v; } } restore_{in_region_, region_was};
        float* dvec = alloc((long)B * H * T);
        s2st_attn_args fb = fa;
        fb.doh = ghalf_of(o);
        Part bp[2];
        const int nb = chain_parts(B * T, bp);
        // q / k / v are column blocks of plain projections: their gradients are only ever read as bf16
        // GEMM operands (+ bias column sums), so the kernels emit exactly that and no fp32 gradient
        const bool gf = use_attn_gfuse && io3.qt->lin_plain && io3.kt->lin_plain && io3.vt->lin_plain &&
                        !io3.qt->g && !io3.kt->g && !io3.vt->g;
        if (gf) {
          for (Ten* t : {io3.qt, io3.kt, io3.vt})
            if (!t->gpre_h) t->gpre_h = alloc_h(t->n());
          fb.dqh = io3.qt->gpre_h + io3.qoff; fb.dkh = io3.kt->gpre_h + io3.koff; fb.dvh = io3.vt->gpre_h + io3.voff;
        } else {
          bool aq, ak, av;
          float* gq = gradbuf(io3.qt, aq);
          float* gk = gradbuf(io3.kt, ak);
          float* gv = gradbuf(io3.vt, av);
          (void)aq; (void)ak; (void)av;  // disjoint column blocks, each written exactly once
          fb.dq = gq + io3.qoff; fb.dk = gk + io3.koff; fb.dv = gv + io3.voff;
        }
        // (dK,dV and dQ are independent, but joining the second stream here would also wait for its
        // backlog of weight-gradient GEMMs: measured slower, so both stay on the data-path stream)
        // Bias gradients of the projections in the fused form.  S2ST_ATTN_GFUSE=1: out of the attention kernels' fp32
        // accumulators BEFORE they are rounded to bf16, as per-(block, wave) partial sums folded in a fixed order (no
        // atomics, no pass over the rounded copies; a key bias's mathematically zero gradient stays ~0 and repeats);
        // =2: the same sums as fp32 atomics per head column (contention: slow); =3: column sums of the rounded bf16
        // copies (round 1's form: a rounding residue of ~1e-5 instead of ~0)
        const bool gf_db = gf && attn_gfuse_mode != 3;
        float* dbp[2] = {nullptr, nullptr};
        if (gf_db) {
          if (io3.qt->act_bias >= 0) fb.dbq = G + io3.qt->act_bias + io3.qoff;
          if (io3.kt->act_bias >= 0) fb.dbk = G + io3.kt->act_bias + io3.koff;
          if (io3.vt->act_bias >= 0) fb.dbv = G + io3.vt->act_bias + io3.voff;
          if (attn_gfuse_mode != 2)
            for (int ci = 0; ci < nb; ++ci) dbp[ci] = alloc(s2st_flash_attn_db_scratch_floats(&fb));  // (sized for the whole batch)
        }
        for (int ci = 0; ci < nb; ++ci) {
          const int b0 = bp[ci].r0 / T, nbat = bp[ci].nr / T;
          const s2st_attn_args a = chain_args(fb, b0, nbat, bp[ci].salt);
          if (live()) chk(s2st_flash_attn_bwd(&a, o->g + (long)b0 * T * C, dvec + (long)b0 * H * T, bp[ci].st, 0, dbp[ci]));
          if (dbp[ci] && live()) {
            // parameter gradients: the partials join the segment's batched fold (flush_lnfold, second stream)
            int sq = 0, sk = 0;
            s2st_flash_attn_db_layout(&a, &sq, &sk);
            const int Cm = a.H * a.dh;
            if (pending_lnfold.n + 3 > S2ST_LNFOLD_MAX) flush_lnfold();
            if (a.dbq) chk(s2st_fold_add(pending_lnfold, dbp[ci], sq, Cm, 1, a.dbq, nullptr, nullptr));
            if (a.dbk) chk(s2st_fold_add(pending_lnfold, dbp[ci] + (long)sq * Cm, sk, Cm, 1, a.dbk, nullptr, nullptr));
            if (a.dbv) chk(s2st_fold_add(pending_lnfold, dbp[ci] + (long)(sq + sk) * Cm, sk, Cm, 1, a.dbv, nullptr, nullptr));
          }
        }
        if (gf && !gf_db) {
          // projection bias gradients = column sums of the bf16 gradients: parameter gradients only, so
          // on the second stream (atomics from inside the attention kernels contend on H*dh addresses)
          hipStream_t bs = live() ? fork_side() : st_;
          Ten* seen[3] = {nullptr, nullptr, nullptr};
          int ns = 0;
          for (Ten* t : {io3.qt, io3.kt, io3.vt}) {
            bool dup = false;
            for (int i = 0; i < ns; ++i) dup = dup || seen[i] == t;
            if (dup || t->act_bias < 0) continue;
            seen[ns++] = t;
            // (fixed-order sums, no atomics: the key bias's gradient is pure rounding noise and must repeat)
            float* part = alloc(s2st_colsum_bf16_scratch_floats(t->rows, t->cols));
            if (live()) chk(s2st_colsum_bf16_ordered(t->gpre_h, t->cols, t->rows, t->cols, G + t->act_bias, part, bs));
          }
        }
      });
      set_aware();
      return o;
    }
    if (live()) sync_chains();  // (the unfused path below works on whole-batch score tensors)
    float* p = alloc((long)B * H * T * ld);
    float* pd = drop_p > 0.f ? alloc((long)B * H * T * ld) : p;
    bf16raw* pdh = fm ? alloc_h((long)B * H * T * ld) : nullptr;  // bf16 dropout(p): the P*V / dV operand
    const uint64_t sd = drop_p > 0.f ? next_seed(S2ST_SITE_ATTN, drop_p, B, H, T, S, ld) : 0;
    const float scaling = 1.0f / sqrtf((float)dh);
    const int prec = c.precise;
    // operand views: (fp32 base, bf16 base) + element offset; strides are the same in both
    struct View { const float* f; const bf16raw* h; };
    auto bgemm = [=](View A, int akm, long ald, long azo, View Bv, int bkm, long bld, long bzo, float* Cp,
                     long cld, long czo, long czi, long azi, long bzi, int M, int N, int K, float alpha,
                     bf16raw* Ch = nullptr) {
      GemmArgs g{};
      if (fm) {
        g.A = akm ? gemm_rowmajor(A.h, ald) : gemm_colmajor(A.h, ald);
        g.B = bkm ? gemm_rowmajor(Bv.h, bld) : gemm_colmajor(Bv.h, bld);
      } else {
        g.A = akm ? gemm_rowmajor(A.f, ald) : gemm_colmajor(A.f, ald);
        g.B = bkm ? gemm_rowmajor(Bv.f, bld) : gemm_colmajor(Bv.f, bld);
      }
      g.A.zo = azo; g.A.zi = azi;
      g.B.zo = bzo; g.B.zi = bzi;
      g.C = gemm_out(Cp, cld);
      g.C.zo = czo; g.C.zi = czi;
      g.C.h = Ch;
      g.ep = gemm_epi_default();
      g.ep.alpha = alpha;
      g.M = M; g.N = N; g.K = K; g.batch = B * H; g.zdiv = H; g.precise = prec;
      chk(s2st_gemm(g, st_));
    };
    const long pzo = (long)H * T * ld, pzi = (long)T * ld;
    // q / k / v column blocks of their holders (row stride == hld: every projection width is % 8)
    const bf16raw *qh = nullptr, *kh = nullptr, *vh = nullptr;
    if (fm) { qh = half_of(io.qt); kh = half_of(io.kt); vh = half_of(io.vt); }
    const View Vq{io.qt->d + io.qoff, fm ? qh + io.qoff : nullptr};
    const View Vk{io.kt->d + io.koff, fm ? kh + io.koff : nullptr};
    const View Vv{io.vt->d + io.voff, fm ? vh + io.voff : nullptr};
    if (live()) {
      // scores = (q * dh^-0.5) k^T      (multihead_attention.py:224, 332)
      bgemm(Vq, 1, io.ldq, (long)T * io.ldq, Vk, 1, io.ldk, (long)S * io.ldk, p, ld, pzo, pzi, dh, dh, T, S, dh,
            scaling);
      chk(s2st_softmax_fwd(p, p, drop_p > 0.f ? pd : nullptr, klen, B, H, T, S, ld, causal, drop_p, sd, st_, pdh));
      // o = dropout(p) v                 (:367)
      bgemm(View{pd, pdh}, 1, ld, pzo, Vv, 0, io.ldv, (long)S * io.ldv, o->d, C, (long)T * C, dh, pzi, dh, T, dh,
            S, 1.f, o->h);
      if (attn_mean_out) chk(s2st_attn_headmean(p, attn_mean_out, B, H, T, S, ld, st_));
    }
    AttnIO io2 = io;
    tape.push_back([=]() {
      if (!o->g) return;
      // q/k/v gradients are written into column blocks of their holders' gradient buffers
      bool aq, ak, av;
      float* gq = gradbuf(io2.qt, aq);
      float* gk = gradbuf(io2.kt, ak);
      float* gv = gradbuf(io2.vt, av);
      (void)aq; (void)ak; (void)av;  // column blocks are disjoint and written exactly once
      float* dp = alloc((long)B * H * T * ld);
      bf16raw* dsh = fm ? alloc_h((long)B * H * T * ld) : nullptr;
      const bf16raw* doh = fm ? ghalf_of(o) : nullptr;
      if (!live()) return;
      const View Vdo{o->g, doh};
      // dPd = dO V^T
      bgemm(Vdo, 1, C, (long)T * C, Vv, 1, io2.ldv, (long)S * io2.ldv, dp, ld, pzo, pzi, dh, dh, T, S, dh, 1.f);
      // dV = Pd^T dO
      bgemm(View{pd, pdh}, 0, ld, pzo, Vdo, 0, C, (long)T * C, gv + io2.voff, io2.ldv, (long)S * io2.ldv, dh, pzi,
            dh, S, dh, T, 1.f);
      chk(s2st_softmax_bwd(p, dp, dp, B, H, T, S, ld, drop_p, sd, st_, dsh));
      // dQ = scaling * dS K ; dK = scaling * dS^T Q
      bgemm(View{dp, dsh}, 1, ld, pzo, Vk, 0, io2.ldk, (long)S * io2.ldk, gq + io2.qoff, io2.ldq,
            (long)T * io2.ldq, dh, pzi, dh, T, dh, S, scaling);
      bgemm(View{dp, dsh}, 0, ld, pzo, Vq, 0, io2.ldq, (long)T * io2.ldq, gk + io2.koff, io2.ldk,
            (long)S * io2.ldk, dh, pzi, dh, S, dh, T, scaling);
    });
    return o;
  }

  Ten* self_attn_block(Ten* x, const AttnP& a, int B, int T, int H, const int* klen, int causal,
                       Ten* resid) {
    const int C = x->cols;
    Ten* kvq = linear(x, a.kvq_w, a.kvq_b, 3 * C, C, 0, 0.f, nullptr, nullptr, true);
    AttnIO io{kvq, 2 * C, 3 * C, kvq, 0, 3 * C, kvq, C, 3 * C};
    Ten* o = attention(io, B, T, T, H, C / H, klen, causal, bt.training ? c.attn_dropout : 0.f, nullptr);
    return linear(o, a.out_w, a.out_b, C, C, 0, bt.training ? c.dropout : 0.f, resid);
  }
  Ten* cross_kv(Ten* encx, const XAttnP& a, int C) {
    return linear(encx, a.kv_w, a.kv_b, 2 * C, encx->cols, 0, 0.f, nullptr, nullptr, true);
  }
  Ten* cross_attn_block(Ten* x, Ten* encx, const XAttnP& a, int B, int T, int S, int H,
                        const int* klen, Ten* resid, float* attn_mean_out, Ten* kv_pre = nullptr) {
    const int C = x->cols;
    Ten* q = linear(x, a.q_w, a.q_b, C, C, 0, 0.f, nullptr, nullptr, true);
    Ten* kv = kv_pre ? kv_pre : cross_kv(encx, a, C);
    if (kv_pre && kv_wait_) {  // first consumer of the projections issued on the second stream
      wait_traced(st_, ev_kv_, "cross-attention K|V projections");
      chains_wait(ev_kv_);
      kv_wait_ = false;
    }
    AttnIO io{q, 0, C, kv, 0, 2 * C, kv, C, 2 * C};
    Ten* o = attention(io, B, T, S, H, C / H, klen, 0, bt.training ? c.attn_dropout : 0.f, attn_mean_out);
    return linear(o, a.out_w, a.out_b, C, C, 0, bt.training ? c.dropout : 0.f, resid);
  }
  Ten* ffn_block(Ten* x, const LinP& fc1, const LinP& fc2, Ten* resid) {
    // the hidden activation only feeds fc2's GEMM: bf16 copy only (training needs the ReLU form of it for the fused
    // backward; the frozen HuBERT layers, GELU, are forward-only)
    Ten* h = linear(x, fc1.w, fc1.b, fc1.N, fc1.K, ffn_act, bt.training ? c.act_dropout : 0.f, nullptr, nullptr,
                    ffn_act == 1 || is_hubert);
    return linear(h, fc2.w, fc2.b, fc2.N, fc2.K, 0, bt.training ? c.dropout : 0.f, resid);
  }
  Ten* enc_layer(Ten* x, const EncLayerP& l, int B, int T) {
    const int H = c.enc_heads;
    if (c.enc_pre_ln) {  // the normalised activations only feed GEMMs: bf16 copy only
      x = self_attn_block(layernorm(x, l.ln1, nullptr, true), l.sa, B, T, H, bt.enc_lens, 0, x);
      return ffn_block(layernorm(x, l.ln2, nullptr, true), l.fc1, l.fc2, x);
    }
    x = layernorm(self_attn_block(x, l.sa, B, T, H, bt.enc_lens, 0, x), l.ln1);
    return layernorm(ffn_block(x, l.fc1, l.fc2, x), l.ln2);
  }
  Ten* dec_layer(Ten* x, Ten* encx, const DecLayerP& l, int B, int T, int S, int H, bool pre_ln,
                 const int* self_klen, float* attn_mean_out, Ten* kv_pre = nullptr) {
    if (pre_ln) {
      x = self_attn_block(layernorm(x, l.ln1, nullptr, true), l.sa, B, T, H, self_klen, 1, x);
      x = cross_attn_block(layernorm(x, l.ln2, nullptr, true), encx, l.xa, B, T, S, H, bt.enc_lens, x, attn_mean_out, kv_pre);
      return ffn_block(layernorm(x, l.ln3, nullptr, true), l.fc1, l.fc2, x);
    }
    x = layernorm(self_attn_block(x, l.sa, B, T, H, self_klen, 1, x), l.ln1);
    x = layernorm(cross_attn_block(x, encx, l.xa, B, T, S, H, bt.enc_lens, x, attn_mean_out, kv_pre), l.ln2);
    return layernorm(ffn_block(x, l.fc1, l.fc2, x), l.ln3);
  }

  // conv over a halo-padded input.  xh: [B][Tin + 2*pad][I] (zeros in the halo); returns z [B*Tout][O].
  // The backward needs dz both plain (weight gradient / bias) and as a halo-padded, for
  // stride 2 zero-stuffed, image (data gradient as a stride-1 correlation with flipped taps).
  struct ConvIn { float* xh; Ten* src; int Tin; const bf16raw* xhh; };  // src: plain tensor whose grad we produce (or null); xhh: bf16 twin of xh
  struct ConvW { float *wf, *wd, *dwf; const bf16raw *wfh, *wdh; };
  Ten* conv(const ConvIn& in, const ConvP& p, int B, int stride, const ConvW& cw) {
    float *wf = cw.wf, *wd = cw.wd, *dwf = cw.dwf;
    const bf16raw *wfh = cw.wfh, *wdh = cw.wdh;
    const bool fm = fast();
    const int pad = p.Kw / 2;
    const int Tin = in.Tin, Tout = (Tin + 2 * pad - p.Kw) / stride + 1;
    const int Th = Tin + 2 * pad;
    Ten* z = newT(B * Tout, p.O);
    touch(p.w + (long)p.O * p.I * p.Kw);
    touch(p.b + p.O);
    if (live()) {
      GemmArgs g{};
      g.A = fm ? gemm_rowmajor(in.xhh, (long)stride * p.I) : gemm_rowmajor(in.xh, (long)stride * p.I);
      g.A.sp.per = Tout; g.A.sp.bs = (long)Th * p.I;
      g.B = fm ? gemm_rowmajor(wfh, (long)p.Kw * p.I) : gemm_rowmajor(wf, (long)p.Kw * p.I);
      g.C = gemm_out(z->d, p.O);
      g.ep = gemm_epi_default();
      g.ep.bias = P + p.b;
      g.M = B * Tout; g.N = p.O; g.K = p.Kw * p.I; g.batch = 1; g.zdiv = 1; g.precise = c.precise;
      chk(s2st_gemm(g, st_));
    }
    ConvP pp = p;
    ConvIn in2 = in;
    tape.push_back([=]() {
      if (!z->g) return;
      const int M = B * Tout;
      const bf16raw* dzh = fm ? ghalf_of(z) : nullptr;
      // (fixed-order bias sums: a convolution bias in front of BatchNorm has a mathematically zero gradient)
      float* cpart = ordered_sums ? alloc(s2st_colsum_scratch_floats(M, pp.O)) : nullptr;
      const bool no_dgrad = !(in2.src && in2.src->needs_grad);
      // dz placed at rows pad + stride*t of a zeroed [B][Tin + 2 pad][O] image: the data gradient's operand (a stride-1
      // correlation with flipped taps) -- fast mode builds it directly in bf16 from dz's bf16 twin (no fp32 image, no cast pass).
      // (Round 5 also expressed the WEIGHT gradient over whole halo-image rows so that it could join the grouped LDS-DMA
      //  launch: built, tested, the second stream did 0.11 ms less and the step got 0.07 ms SLOWER -- a grouped workgroup holds
      //  128 KB of LDS where the split-row kernel leaves room for the data path's workgroups; profiles/r05_conv_wgrad_ab.txt.
      //  Removed in round 6.)
      const bool direct = fm && pp.O % 8 == 0;
      const bool need_img = !no_dgrad;
      float* up = (need_img && !direct) ? alloc((long)B * Th * pp.O, true) : nullptr;
      bf16raw* upd = (need_img && direct) ? alloc_h((long)B * Th * pp.O) : nullptr;
      if (need_img && live()) {
        Split xs{(long)pp.O, 0, 0, 0};
        Split ys{(long)stride * pp.O, (long)Th * pp.O, Tout, 0};
        if (direct) {
          chk(s2st_halo_image_bf16(dzh, z->hld(), upd, B, Tout, Th, pp.O, pad, stride, st_));
        } else {
          chk(s2st_copy_rows(z->g, xs, up + (long)pad * pp.O, ys, M, pp.O, st_));
        }
      }
      if (live()) {
        // parameter gradients only: on the second stream, next to the data-gradient chain -- except for a convolution
        // whose input needs no gradient (the model's first one = the LAST closure of the backward): no data-gradient
        // chain is left, the data-path stream would only wait, so it takes the product and the second stream the bias sum
        hipStream_t side_st = fm ? fork_side() : st_;
        hipStream_t ws_st = (fm && !no_dgrad) ? side_st : st_;
        {
          GemmArgs g{};  // dWf[O][(j,c)] += sum_(b,t) dz[(b,t)][o] * xh[b][t*stride + j][c]
          g.A = fm ? gemm_colmajor(dzh, z->hld()) : gemm_colmajor(z->g, pp.O);
          g.B = fm ? gemm_colmajor(in2.xhh, (long)stride * pp.I) : gemm_colmajor(in2.xh, (long)stride * pp.I);
          g.B.sp.per = Tout; g.B.sp.bs = (long)Th * pp.I;
          g.C = gemm_out(dwf, (long)pp.Kw * pp.I);
          g.ep = gemm_epi_default();
          g.ep.accumulate = 1;
          g.ws = ws_for(ws_st); g.ws_floats = skws_n;
          g.M = pp.O; g.N = pp.Kw * pp.I; g.K = M; g.batch = 1; g.zdiv = 1; g.precise = c.precise;
          chk(s2st_gemm(g, ws_st));
        }
        {
          int slabs = 0;
          chk(s2st_colsum(z->g, pp.O, M, pp.O, G + pp.b, 1, side_st, cpart, cpart ? &slabs : nullptr));
          if (cpart) add_fold(cpart, slabs, pp.O, G + pp.b);
        }
        chk(s2st_conv_w_unpermute_acc(dwf, G + pp.w, pp.O, pp.I, pp.Kw, ws_st, 1));
      }
      if (!no_dgrad) {
        bool acc;
        float* dx = gradbuf(in2.src, acc);
        const bf16raw* uph = direct ? upd : (fm ? cast_buf(up, (long)B * Th * pp.O) : nullptr);
        if (live()) {
          GemmArgs g{};  // dx[(b,u)][c] = sum_(j',o) up[b][u + j'][o] * Wd[c][j'][o]
          g.A = fm ? gemm_rowmajor(uph, pp.O) : gemm_rowmajor(up, pp.O);
          g.A.sp.per = Tin; g.A.sp.bs = (long)Th * pp.O;
          g.B = fm ? gemm_rowmajor(wdh, (long)pp.Kw * pp.O) : gemm_rowmajor(wd, (long)pp.Kw * pp.O);
          g.C = gemm_out(dx, pp.I);
          g.ep = gemm_epi_default();
          g.ep.accumulate = acc ? 1 : 0;
          g.ws = ws_for(st_); g.ws_floats = skws_n;  // split-K through slabs (fixed order), not atomics
          g.M = B * Tin; g.N = pp.I; g.K = pp.Kw * pp.O; g.batch = 1; g.zdiv = 1; g.precise = c.precise;
          chk(s2st_gemm(g, st_));
        }
      }
    });
    return z;
  }

  // GLU of z [rows][2C] into a halo-padded image [B][T + 2 pad][C]; returns the plain-gradient
  // holder for the image (its grad is [rows][C] plain)
  Ten* glu_to(Ten* z, float* y, Split ysp, int Cc) {
    Ten* holder = newT(z->rows, Cc, y);  // d points at the (possibly halo) image; only g is used plainly
    if (live()) chk(s2st_glu_fwd(z->d, y, ysp, z->rows, Cc, st_));
    tape.push_back([=]() {
      if (!holder->g) return;
      bool acc;
      float* dz = gradbuf(z, acc);
      (void)acc;  // single consumer
      Split ds{(long)Cc, 0, 0, 0}, das{(long)2 * Cc, 0, 0, 0};
      if (fast() && !z->gh) z->gh = alloc_h((long)z->rows * z->hld());  // the conv backward's GEMM operand
      if (live()) chk(s2st_glu_bwd(z->d, holder->g, ds, dz, das, z->rows, Cc, st_, z->gh, z->hld()));
    });
    return holder;
  }

  // spk_off >= 0: + the utterance's speaker-embedding row at every one of its T positions (before the dropout)
  Ten* add_pe(Ten* x, const int* pos, const float* table, float scale, long alpha_off, float drop_p, long spk_off = -1,
              int T = 0) {
    Ten* y = newT(x->rows, x->cols);
    const uint64_t sd = drop_p > 0.f ? next_seed(S2ST_SITE_ROWS, drop_p, x->rows, x->cols) : 0;
    if (alpha_off >= 0) touch(alpha_off + 1);
    const bool spk = spk_off >= 0 && bt.speaker != nullptr;
    if (spk) touch_spk(spk_off + (long)c.n_speakers * x->cols);
    const long* spk_ids = (const long*)bt.speaker;
    const int Bn = T > 0 ? x->rows / T : 0;
    if (live())
      chk(s2st_add_pe(x->d, y->d, pos, table, x->rows, x->cols, scale, alpha_off >= 0 ? P + alpha_off : nullptr,
                      drop_p, sd, st_, spk ? spk_tab(spk_off) : nullptr, spk ? spk_ids : nullptr, T));
    tape.push_back([=]() {
      if (!y->g) return;
      if (spk && !c.spk_frozen && live())
        chk(s2st_speaker_bwd(y->g, spk_ids, Bn, T, T, x->cols, c.n_speakers, drop_p, sd, G + spk_off, st_));
      float* apart = (alpha_off >= 0 && ordered_sums) ? alloc(1024) : nullptr;
      if (alpha_off >= 0 && live()) {
        int np = 0;
        chk(s2st_pe_alpha_bwd(y->g, pos, table, x->rows, x->cols, drop_p, sd, G + alpha_off, st_, apart, apart ? &np : nullptr));
        if (apart) add_fold(apart, np, 1, G + alpha_off);
      }
      if (x->needs_grad) {
        bool acc;
        float* dx = gradbuf(x, acc);
        if (live()) chk(s2st_dropout(y->g, dx, x->n(), scale, drop_p, sd, acc ? 1 : 0, st_));
      }
    });
    return y;
  }

  Ten* aux_decoder(const AuxP& a, Ten* tap, const long* prev_tok, const int* pos, const int* lens, int B,
                   int L, const float* pe, float* logits_out) {
    const int E = bt.E;
    Ten* emb = newT(B * L, a.in_dim);
    const float scale = c.no_scale_embedding ? 1.f : sqrtf((float)a.d);
    touch(a.embed + (long)a.V * a.in_dim);
    if (live()) chk(s2st_embed_fwd(prev_tok, P + a.embed, emb->d, B * L, a.in_dim, scale, st_));
    long embed_off = a.embed;
    int in_dim = a.in_dim;
    const int vocab = a.V;
    tape.push_back([=]() {
      if (!emb->g) return;
      if (live()) chk(s2st_embed_bwd(prev_tok, emb->g, G + embed_off, B * L, in_dim, scale, 1, st_, ordered_sums ? vocab : 0));
    });
    Ten* x = emb;
    if (a.proj_in >= 0) x = linear(x, a.proj_in, -1, a.d, a.in_dim);
    const char* who = &a == &asr ? "asr" : (&a == &st ? "st" : "s2t");
    snprintf(site_ctx, sizeof site_ctx, "%s.pe", who);
    x = add_pe(x, pos, pe, 1.f, -1, bt.training ? c.dropout : 0.f);
    for (int i = 0; i < a.layers; ++i) {
      snprintf(site_ctx, sizeof site_ctx, "%s.L%d", who, i);
      x = dec_layer(x, tap, a.L[i], B, L, E, c.dec_heads, c.dec_pre_ln != 0, lens, nullptr);
    }
    site_ctx[0] = 0;
    if (a.has_ln) x = layernorm(x, a.ln);
    if (a.proj_out >= 0) x = linear(x, a.proj_out, -1, a.out_dim, a.d);
    return linear(x, a.out_proj, -1, a.V, a.out_dim, 0, 0.f, nullptr, logits_out);
  }


  // ------------------------------------------------------------------------------------
  // incremental decoding (fairseq/speech_generator_for_s2st.py:46-110; s2st_transformer.py:369-456 with
  // incremental_state; transformer_layer.py:301-446; multihead_attention.py:194-385 incremental path)
  // k_new / v_new (self-attention): this step's key / value rows [B][ld_new]; every (b, h) workgroup writes its head slice
  // to row `pos_new` of the caches before it attends (the two copy launches per layer of rounds 1 - 3 are gone)
  Ten* dec_attn(Ten* qt, int qoff, float* K, float* V, long ldk, long kbs, const int* klen, int nkeys,
                int H, float* attn_mean, int S, const float* k_new = nullptr, const float* v_new = nullptr, long ld_new = 0,
                int pos_new = 0, int kv_bf16 = 0, const int* step_ptr = nullptr, int dim = 0, int rows = 0) {
    const int Cd = dim > 0 ? dim : c.dec_dim, B = rows > 0 ? rows : dec_st.B;  // (dim / rows: an aux text decoder's, below)
    Ten* o = newT(B, Cd);
    if (live())
      chk(s2st_decode_attn(qt->d + qoff, qt->cols, K, V, ldk, kbs, klen, nkeys, B, H, Cd / H,
                           1.0f / sqrtf((float)(Cd / H)), o->d, Cd, attn_mean, S, st_, k_new, v_new, ld_new, pos_new, kv_bf16,
                           step_ptr));
    return o;
  }

  // ---- incremental decoding of an aux ASR / ST text decoder (beam search: fairseq/sequence_generator.py:189-571 carries an
  //      incremental_state and calls reorder_incremental_state with the surviving beams' indices every step;
  //      fairseq/modules/multihead_attention.py:261-299 appends the step's key / value rows to the cached ones and keeps the
  //      static encoder keys / values).  State = a caller-owned buffer: TWO copies of the self-attention caches
  //      [layer][K | V][Bb][maxT][d] (a step that reorders gathers the valid rows of every hypothesis from one copy into the
  //      other) and the per-layer encoder K | V projections [layer][Bb * E][2 d] of the head's encoder tap.  A hypothesis then
  //      costs O(L) per step instead of a re-run of the decoder on its whole prefix. -----------------------------------------
  struct AuxInc {
    float* base = nullptr; int Bb = 0, E = 0, maxT = 0, cur = 0; const int* enc_lens = nullptr;
  } aux_inc[2];
  static long aux_inc_half(const AuxP& a, int Bb, int maxT) { return (long)a.layers * 2 * Bb * maxT * a.d; }
  float* aux_selfK(const AuxP& a, const AuxInc& S, int l, int buf) const {
    return S.base + (long)buf * aux_inc_half(a, S.Bb, S.maxT) + (long)l * 2 * S.Bb * S.maxT * a.d;
  }
  float* aux_crossKV(const AuxP& a, const AuxInc& S, int l) const {
    return S.base + 2 * aux_inc_half(a, S.Bb, S.maxT) + (long)l * S.Bb * S.E * 2 * a.d;
  }
  int aux_inc_begin(const AuxP& a, AuxInc& S, Ten* tap) {
    bt.training = 0;
    for (int l = 0; l < a.layers; ++l) {
      const XAttnP& xa = a.L[l].xa;
      linear(tap, xa.kv_w, xa.kv_b, 2 * a.d, tap->cols, 0, 0.f, nullptr, aux_crossKV(a, S, l));
    }
    S.cur = 0;
    return err;
  }
  // tokens [Bb]: the hypotheses' LAST tokens; reorder [Bb] (or null): hypothesis b continues old hypothesis reorder[b]
  // (fairseq's reorder_incremental_state); pos [Bb]: the tokens' positions (step + 2: prefixes hold no padding)
  int aux_inc_step(const AuxP& a, AuxInc& S, int step, const long* tokens, const int* reorder, const int* pos, const float* pe,
                   float* logits_out) {
    const int Bb = S.Bb, d = a.d, H = c.dec_heads, maxT = S.maxT, E = S.E;
    if (!S.base || step < 0 || step >= maxT) return S2ST_ERR_ARG;
    bt.training = 0;
    const bool pre = c.dec_pre_ln != 0;
    if (reorder && step > 0) {
      if (live())
        chk(s2st_cache_reorder(S.base + (long)S.cur * aux_inc_half(a, Bb, maxT), S.base + (long)(1 - S.cur) * aux_inc_half(a, Bb, maxT),
                               reorder, 2 * a.layers, Bb, (long)maxT * d, (long)step * d, st_));
      S.cur ^= 1;
    }
    Ten* emb = newT(Bb, a.in_dim);
    emb->needs_grad = false;
    const float scale = c.no_scale_embedding ? 1.f : sqrtf((float)a.d);
    touch(a.embed + (long)a.V * a.in_dim);
    if (live()) chk(s2st_embed_fwd(tokens, P + a.embed, emb->d, Bb, a.in_dim, scale, st_));
    Ten* x = emb;
    if (a.proj_in >= 0) x = linear(x, a.proj_in, -1, d, a.in_dim);
    x = add_pe(x, pos, pe, 1.f, -1, 0.f);
    for (int l = 0; l < a.layers; ++l) {
      const DecLayerP& L = a.L[l];
      float* Kc = aux_selfK(a, S, l, S.cur);
      float* Vc = Kc + (long)Bb * maxT * d;
      Ten* kvq = pre ? ln_linear(x, L.ln1, L.sa.kvq_w, L.sa.kvq_b, 3 * d, d) : linear(x, L.sa.kvq_w, L.sa.kvq_b, 3 * d, d);
      Ten* o = dec_attn(kvq, 2 * d, Kc, Vc, d, (long)maxT * d, nullptr, step + 1, H, nullptr, 0, kvq->d, kvq->d + d, 3 * d, step, 0,
                        nullptr, d, Bb);
      x = linear(o, L.sa.out_w, L.sa.out_b, d, d, 0, 0.f, x);
      if (!pre) x = layernorm(x, L.ln1);
      Ten* q = pre ? ln_linear(x, L.ln2, L.xa.q_w, L.xa.q_b, d, d) : linear(x, L.xa.q_w, L.xa.q_b, d, d);
      o = dec_attn(q, 0, aux_crossKV(a, S, l), aux_crossKV(a, S, l) + d, 2 * d, (long)E * 2 * d, S.enc_lens, E, H, nullptr, E, nullptr,
                   nullptr, 0, 0, 0, nullptr, d, Bb);
      x = linear(o, L.xa.out_w, L.xa.out_b, d, d, 0, 0.f, x);
      if (!pre) x = layernorm(x, L.ln2);
      if (pre) {
        Ten* hdn = ln_linear(x, L.ln3, L.fc1.w, L.fc1.b, L.fc1.N, L.fc1.K, ffn_act);
        x = linear(hdn, L.fc2.w, L.fc2.b, L.fc2.N, L.fc2.K, 0, 0.f, x);
      } else {
        x = layernorm(ffn_block(x, L.fc1, L.fc2, x), L.ln3);
      }
    }
    if (a.has_ln) x = layernorm(x, a.ln);
    if (a.proj_out >= 0) x = linear(x, a.proj_out, -1, a.out_dim, d);
    linear(x, a.out_proj, -1, a.V, a.out_dim, 0, 0.f, nullptr, logits_out);
    return err;
  }

  int decode_step(int step, const float* prev, const int* pos, const int* self_klen, uint64_t sd, float* feat_out,
                  float* eos_prob, float* attn_out) {
    const int B = dec_st.B, Cd = c.dec_dim, H = c.dec_heads, E = dec_st.E, maxT = dec_st.maxT;
    if (!dec_st.base || step < 0 || step >= maxT) return S2ST_ERR_ARG;
    bt.training = 0;
    seed = sd;
    const bool pre = c.dec_pre_ln != 0;
    Ten* x = newT(B, c.out_dim, const_cast<float*>(prev));
    x->needs_grad = false;
    if (dec_spk >= 0 && bt.speaker) {
      // the reference's decoder replaces prev_output_tokens[:, 0] with the speaker row and keeps [:, 1:]
      // (s2st_transformer.py:441-444); its generator hands over ONE frame per step (speech_generator_for_s2st.py:84-99),
      // so during incremental decoding EVERY step's input is the speaker row and the fed-back feature is dropped.
      // Reproduced as is (results identical to the reference's).
      Ten* sp = newT(B, c.out_dim);
      touch_spk(dec_spk + (long)c.n_speakers * c.out_dim);  // (an overlapped optimizer update may still be writing the table)
      if (live()) chk(s2st_embed_fwd((const long*)bt.speaker, spk_tab(dec_spk), sp->d, B, c.out_dim, 1.f, st_));
      sp->needs_grad = false;
      x = sp;
    }
    // Prenet: dropout is ALWAYS on (tacotron2.py:95-98), also at inference
    for (int i = 0; i < c.prenet_layers; ++i)
      x = linear(x, prenet[i].w, prenet[i].b, prenet[i].N, prenet[i].K, 1, c.prenet_dropout);
    // (every utterance is at position step + 2 in the incremental path: x + alpha * PE[pos] is ONE row for the whole
    // batch, taken from the alpha-scaled table decode_begin prepared -- added in the projection's epilogue on the skinny
    // path, by the position kernel otherwise)
    const float* pe_row = dec_st.pe_alpha ? (replay_ ? replay_->pe_cur : dec_st.pe_alpha + (long)(step + 2) * Cd) : nullptr;
    const bool pe_fused = pe_row && fast() && use_skinny && B <= S2ST_SKINNY_MAX_ROWS && c.prenet_dim % 32 == 0;
    if (replay_ && !pe_fused) return S2ST_ERR_SHAPE;  // (the position kernel takes this step's rows: not replayable)
    x = linear(x, prenet.back().w, prenet.back().b, Cd, c.prenet_dim, 0, 0.f, nullptr, nullptr, false, pe_fused ? pe_row : nullptr);
    if (!pe_fused) x = add_pe(x, pos, dec_st.pe_dec, 1.f, pos_alpha, 0.f);
    for (int l = 0; l < c.dec_layers; ++l) {
      const DecLayerP& L = dec[l];
      // self-attention over the cached keys / values 0..step
      Ten* kvq = pre ? ln_linear(x, L.ln1, L.sa.kvq_w, L.sa.kvq_b, 3 * Cd, Cd) : linear(x, L.sa.kvq_w, L.sa.kvq_b, 3 * Cd, Cd);
      // keys >= self_klen[b] are masked: a finished utterance keeps its final length (the reference's
      // cached key padding mask, speech_generator_for_s2st.py:88-89 + multihead_attention.py:268-277)
      // (replay form: the kernel takes keys 0 .. *step and cache row *step; the host-side bound is the whole cache)
      Ten* o = dec_attn(kvq, 2 * Cd, dec_selfK(l), dec_selfV(l), Cd, (long)maxT * Cd, self_klen, replay_ ? maxT : step + 1, H, nullptr, 0,
                        kvq->d, kvq->d + Cd, 3 * Cd, replay_ ? 0 : step, 0, replay_ ? replay_->step : nullptr);
      x = linear(o, L.sa.out_w, L.sa.out_b, Cd, Cd, 0, 0.f, x);
      if (!pre) x = layernorm(x, L.ln1);
      // encoder attention (static keys / values precomputed by decode_begin)
      Ten* q = pre ? ln_linear(x, L.ln2, L.xa.q_w, L.xa.q_b, Cd, Cd) : linear(x, L.xa.q_w, L.xa.q_b, Cd, Cd);
      const bool align = l == c.dec_layers - 1;
      o = dec_attn(q, 0, dec_crossKV(l), dec_crossKV(l) + Cd, 2 * Cd, (long)E * 2 * Cd, dec_st.enc_lens, E, H,
                   align ? attn_out : nullptr, E);
      x = linear(o, L.xa.out_w, L.xa.out_b, Cd, Cd, 0, 0.f, x);
      if (!pre) x = layernorm(x, L.ln2);
      if (pre) {
        Ten* hdn = ln_linear(x, L.ln3, L.fc1.w, L.fc1.b, L.fc1.N, L.fc1.K, ffn_act);
        x = linear(hdn, L.fc2.w, L.fc2.b, L.fc2.N, L.fc2.K, 0, 0.f, x);
      } else {
        x = layernorm(ffn_block(x, L.fc1, L.fc2, x), L.ln3);
      }
    }
    // the stop head's logistic rides in its projection's epilogue on the skinny path (act 3)
    const bool sig_fused = fast() && use_skinny && B <= S2ST_SKINNY_MAX_ROWS && Cd % 64 == 0 && (!has_dec_ln || use_ln_skinny);
    if (replay_ && !sig_fused) return S2ST_ERR_SHAPE;
    Ten* eos;
    if (has_dec_ln) {  // both heads read the normalised state
      ln_linear(x, dec_ln, feat_proj.w, feat_proj.b, c.out_dim, Cd, 0, feat_out);
      eos = ln_linear(x, dec_ln, eos_proj.w, eos_proj.b, 1, Cd, sig_fused ? 3 : 0, sig_fused ? eos_prob : nullptr);
    } else {
      linear(x, feat_proj.w, feat_proj.b, c.out_dim, Cd, 0, 0.f, nullptr, feat_out);
      eos = linear(x, eos_proj.w, eos_proj.b, 1, Cd, sig_fused ? 3 : 0, 0.f, nullptr, sig_fused ? eos_prob : nullptr);
    }
    if (!sig_fused && live()) chk(s2st_sigmoid(eos->d, eos_prob, B, st_));
    return err;
  }

  // conv weight in the GEMM layouts (scratch at the bottom of the workspace)
  ConvW make_conv_scratch(const ConvP& p, bool need_wd, bool tr) {
    ConvW s;
    const bool fm = fast();
    long n = (long)p.O * p.I * p.Kw;
    // (fast mode reads only the bf16 twins of the two layouts: no fp32 copies are made)
    s.wf = fm ? nullptr : alloc(n);
    s.wd = (need_wd && !fm) ? alloc(n) : nullptr;
    s.dwf = alloc(n, tr);
    bf16raw* wfh = fm ? alloc_h((n + 7) / 8 * 8) : nullptr;
    bf16raw* wdh = fm && need_wd ? alloc_h((n + 7) / 8 * 8) : nullptr;
    if (live()) chk(s2st_conv_w_permute(P + p.w, s.wf, s.wd, p.O, p.I, p.Kw, st_, wfh, wdh));
    s.wfh = wfh;
    s.wdh = wdh;
    return s;
  }

  // post-net: 5 x (conv k5 -> BatchNorm -> tanh -> dropout), + residual (tacotron2.py:101-126).
  // Training: batch statistics over ALL B*D rows; eval: running statistics.
  Ten* postnet(Ten* feat, int B, int D, bool tr, std::vector<ConvW>& csp, float* post_out) {
    const bool fm = fast();
    Ten* cur = feat;  // plain holder of the current layer input
    const int pp = c.postnet_k / 2;
    // fast mode: the convolutions read bf16 halo images only -- the first one straight from feat's rows, the others
    // written by the BatchNorm kernel of the layer before (no fp32 images, no memsets, no cast passes)
    float* curh = fm ? nullptr : alloc((long)B * (D + 2 * pp) * c.out_dim, true);
    if (live() && !fm) {
      Split xs{(long)c.out_dim, 0, 0, 0};
      Split ys{(long)c.out_dim, (long)(D + 2 * pp) * c.out_dim, D, 0};
      chk(s2st_copy_rows(feat->d, xs, curh + (long)pp * c.out_dim, ys, B * D, c.out_dim, st_));
    }
    const bf16raw* curhh = fm ? cast_halo(feat->d, B, D, pp, c.out_dim, true) : nullptr;
    Ten* post = nullptr;
    float* bn_tmp = alloc(S2ST_BN_TMP_FLOATS(c.postnet_dim > c.out_dim ? c.postnet_dim : c.out_dim));
    for (int i = 0; i < c.postnet_layers; ++i) {
      const ConvP& pc = post_conv[i];
      const BNP& bn = post_bn[i];
      const bool last = i == c.postnet_layers - 1;
      Ten* z = conv(ConvIn{curh, cur, D, curhh}, pc, B, 1, csp[i]);
      float* mean = alloc(bn.C);
      float* var = alloc(bn.C);
      touch(bn.b + bn.C);
      const float pdrop = tr ? c.postnet_dropout : 0.f;
      const uint64_t sd = pdrop > 0.f ? next_seed(S2ST_SITE_NORM, pdrop, (long)B * D, bn.C) : 0;
      float* nexth = nullptr;
      bf16raw* nexthh = nullptr;
      Ten* out;
      Split osp;
      if (last) {
        out = newT(B * D, bn.C, post_out);
        osp = Split{(long)bn.C, 0, 0, 0};
      } else if (fm) {
        nexthh = alloc_h(((long)B * (D + 2 * pp) * bn.C + 7) / 8 * 8);
        out = newT(B * D, bn.C, nullptr, false);  // (only its gradient is ever used)
        osp = Split{(long)bn.C, 0, 0, 0};
      } else {
        nexth = alloc((long)B * (D + 2 * pp) * bn.C, true);
        out = newT(B * D, bn.C, nexth + (long)pp * bn.C);
        osp = Split{(long)bn.C, (long)(D + 2 * pp) * bn.C, D, 0};
      }
      if (live()) {
        const float *m = BUF + bn.rm, *v = BUF + bn.rv;
        if (tr) {
          chk(s2st_bn_stats(z->d, B * D, bn.C, mean, var, BUF + bn.rm, BUF + bn.rv, 0.1f, bn_tmp, st_));
          m = mean; v = var;
        }
        if (nexthh)
          chk(s2st_bn_apply_img(z->d, m, v, P + bn.g, P + bn.b, nullptr, nexthh, B, D, pp, bn.C, 1e-5f, 1, pdrop, sd, st_));
        else
          chk(s2st_bn_apply(z->d, m, v, P + bn.g, P + bn.b, out->d, osp, last ? feat->d : nullptr, B * D,
                            bn.C, 1e-5f, last ? 0 : 1, pdrop, sd, st_));
      }
      BNP bnp = bn;
      tape.push_back([=]() {
        if (!out->g) return;
        if (last) {  // post = feat + postnet(feat): the residual branch
          bool acc;
          float* df = gradbuf(feat, acc);
          if (live()) chk(s2st_dropout(out->g, df, feat->n(), 1.f, 0.f, 0, acc ? 1 : 0, st_));
        }
        bool acc;
        float* dz = gradbuf(z, acc);
        (void)acc;
        Split ps{(long)bnp.C, 0, 0, 0};
        if (fast() && !z->gh) z->gh = alloc_h((long)z->rows * z->hld());  // the conv backward's GEMM operand
        if (live())
          chk(s2st_bn_bwd(out->g, ps, z->d, mean, var, P + bnp.g, P + bnp.b, dz, ps, G + bnp.g, G + bnp.b,
                          bn_tmp, B * D, bnp.C, 1e-5f, last ? 0 : 1, pdrop, sd, st_, z->gh, z->hld()));
      });
      cur = out;
      curh = nexth;
      curhh = nexthh;
      if (last) post = out;
    }
    return post;
  }

  // t2s encoder prenet: n x (conv k -> BatchNorm -> ReLU -> dropout) over [B][T][C] (t2s_transformer.py:55-66, 86-90).
  // Training: batch statistics over ALL B*T positions (padded ones included: their embedding is the zero row, their
  // conv output the bias + neighbours); eval: running statistics.
  Ten* text_prenet(Ten* emb, int B, int T, bool tr, std::vector<ConvW>& csp) {
    const bool fm = fast();
    const int C = c.enc_dim, pp = c.enc_conv_k / 2;
    Ten* cur = emb;
    float* curh = fm ? nullptr : alloc((long)B * (T + 2 * pp) * C, true);
    if (live() && !fm) {
      Split xs{(long)C, 0, 0, 0};
      Split ys{(long)C, (long)(T + 2 * pp) * C, T, 0};
      chk(s2st_copy_rows(emb->d, xs, curh + (long)pp * C, ys, B * T, C, st_));
    }
    const bf16raw* curhh = fm ? cast_halo(emb->d, B, T, pp, C, true) : nullptr;  // (as in postnet())
    float* bn_tmp = alloc(S2ST_BN_TMP_FLOATS(C));
    const int n = (int)enc_conv.size();
    for (int i = 0; i < n; ++i) {
      const ConvP& pc = enc_conv[i];
      const BNP& bn = enc_bn[i];
      const bool last = i == n - 1;
      Ten* z = conv(ConvIn{curh, cur, T, curhh}, pc, B, 1, csp[i]);
      float* mean = alloc(C);
      float* var = alloc(C);
      touch(bn.b + C);
      const float pdrop = tr ? c.enc_dropout : 0.f;
      const uint64_t sd = pdrop > 0.f ? next_seed(S2ST_SITE_NORM, pdrop, (long)B * T, C) : 0;
      float* nexth = nullptr;
      bf16raw* nexthh = nullptr;
      Ten* out;
      Split osp;
      if (last) {
        out = newT(B * T, C);
        osp = Split{(long)C, 0, 0, 0};
      } else if (fm) {
        nexthh = alloc_h(((long)B * (T + 2 * pp) * C + 7) / 8 * 8);
        out = newT(B * T, C, nullptr, false);
        osp = Split{(long)C, 0, 0, 0};
      } else {
        nexth = alloc((long)B * (T + 2 * pp) * C, true);
        out = newT(B * T, C, nexth + (long)pp * C);
        osp = Split{(long)C, (long)(T + 2 * pp) * C, T, 0};
      }
      if (live()) {
        const float *m = BUF + bn.rm, *v = BUF + bn.rv;
        if (tr) {
          chk(s2st_bn_stats(z->d, B * T, C, mean, var, BUF + bn.rm, BUF + bn.rv, 0.1f, bn_tmp, st_));
          m = mean; v = var;
        }
        if (nexthh)
          chk(s2st_bn_apply_img(z->d, m, v, P + bn.g, P + bn.b, nullptr, nexthh, B, T, pp, C, 1e-5f, 2 /* ReLU */, pdrop, sd, st_));
        else
          chk(s2st_bn_apply(z->d, m, v, P + bn.g, P + bn.b, out->d, osp, nullptr, B * T, C, 1e-5f, 2 /* ReLU */, pdrop, sd, st_));
      }
      BNP bnp = bn;
      tape.push_back([=]() {
        if (!out->g) return;
        bool acc;
        float* dz = gradbuf(z, acc);
        (void)acc;
        Split ps{(long)bnp.C, 0, 0, 0};
        if (fast() && !z->gh) z->gh = alloc_h((long)z->rows * z->hld());
        if (live())
          chk(s2st_bn_bwd(out->g, ps, z->d, mean, var, P + bnp.g, P + bnp.b, dz, ps, G + bnp.g, G + bnp.b, bn_tmp, B * T,
                          bnp.C, 1e-5f, 2, pdrop, sd, st_, z->gh, z->hld()));
      });
      cur = out;
      curh = nexth;
      curhh = nexthh;
    }
    return cur;
  }

  // ------------------------------------------------------------------------------------
  // HuBERT (fairseq/models/hubert/hubert.py:412-461, 518-534; wav2vec2.py:736-905): parameters in
  // GEMM-ready layouts (conv weights [O][k][I], the weight-normed pos_conv as its effective weight
  // [G][E/G][k][E/G]); the host wrapper converts from the reference state_dict layouts.
  void build_params_hubert() {
    int cin = 1;
    for (int i = 0; i < hc.n_conv; ++i) {
      std::string pre = "feature_extractor.conv_layers." + std::to_string(i);
      hp.conv_w[i] = add(pre + ".0.weight", {hc.conv_dim[i], hc.conv_k[i], cin});
      if (i == 0) {
        hp.gn_g = add(pre + ".2.weight", {hc.conv_dim[0]});
        hp.gn_b = add(pre + ".2.bias", {hc.conv_dim[0]});
      }
      cin = hc.conv_dim[i];
    }
    hp.ln = add_ln("layer_norm", cin);
    hp.proj = add_lin("post_extract_proj", hc.embed, cin);
    const int Eg = hc.embed / hc.conv_pos_groups;
    hp.pos_w = add("encoder.pos_conv.0.weight", {hc.conv_pos_groups, Eg, hc.conv_pos, Eg});
    hp.pos_b = add("encoder.pos_conv.0.bias", {hc.embed});
    for (int l = 0; l < hc.layers; ++l) {
      std::string pre = "encoder.layers." + std::to_string(l);
      EncLayerP e;
      e.sa = add_self_attn(pre + ".self_attn", hc.embed);
      e.ln1 = add_ln(pre + ".self_attn_layer_norm", hc.embed);
      e.fc1 = add_lin(pre + ".fc1", hc.ffn, hc.embed);
      e.fc2 = add_lin(pre + ".fc2", hc.embed, hc.ffn);
      e.ln2 = add_ln(pre + ".final_layer_norm", hc.embed);
      hp.L.push_back(e);
    }
    hp.enc_ln = add_ln("encoder.layer_norm", hc.embed);
  }

  int hubert_frames(int n) const {
    for (int i = 0; i < hc.n_conv; ++i) n = n < hc.conv_k[i] ? 0 : (n - hc.conv_k[i]) / hc.conv_stride[i] + 1;
    return n;
  }

  int forward_hubert(const float* wave, const int* frame_lens, int B, int N, float* out) {
    const bool fm = fast();
    if (fm && !PH && !dry) return S2ST_ERR_ARG;
    bt = s2st_batch{};
    bt.B = B;
    bt.training = 0;
    bt.enc_lens = frame_lens;
    skws = nullptr; skws_n = 0; skws_side = nullptr;
    // conv0 (1 -> C0) + GroupNorm(C0, C0) over ALL Tn frames of the padded batch + GELU
    const int C0 = hc.conv_dim[0];
    int Tin = (N - hc.conv_k[0]) / hc.conv_stride[0] + 1;
    if (N < hc.conv_k[0] || Tin <= 0) return S2ST_ERR_SHAPE;
    // fast mode: conv1 only reads the bf16 copy, no fp32 activation is allocated or written
    Ten* a = newT(B * Tin, C0, nullptr, !fm);
    float* stats = alloc(s2st_hubert_conv0_stats_floats(B, Tin, C0));
    if (fm) a->h = alloc_h(a->n());
    if (live())
      chk(s2st_hubert_conv0_gn_gelu(wave, P + hp.conv_w[0], P + hp.gn_g, P + hp.gn_b, a->d, a->h, stats, B, N, Tin, C0,
                                    hc.conv_k[0], hc.conv_stride[0], 1e-5f, st_));
    // conv_i + GELU as GEMMs over the channel-last activations (no padding: windows never cross utterances)
    for (int i = 1; i < hc.n_conv; ++i) {
      const int k = hc.conv_k[i], sd = hc.conv_stride[i], I = hc.conv_dim[i - 1], O = hc.conv_dim[i];
      const int Tout = Tin < k ? 0 : (Tin - k) / sd + 1;
      if (Tout <= 0) return S2ST_ERR_SHAPE;
      Ten* y = newT(B * Tout, O);
      if (fm) y->h = alloc_h(y->n());
      if (live()) {
        GemmArgs g{};
        g.A = fm ? gemm_rowmajor(a->h, (long)sd * I) : gemm_rowmajor(a->d, (long)sd * I);
        g.A.sp.per = Tout; g.A.sp.bs = (long)Tin * I;
        g.B = fm ? gemm_rowmajor(PH + hp.conv_w[i], (long)k * I) : gemm_rowmajor(P + hp.conv_w[i], (long)k * I);
        g.C = gemm_out(y->d, O);
        g.C.h = y->h;
        g.ep = gemm_epi_default();
        g.ep.act = 2;
        g.M = B * Tout; g.N = O; g.K = k * I; g.batch = 1; g.zdiv = 1; g.precise = c.precise;
        chk(s2st_gemm(g, st_));
      }
      a = y;
      Tin = Tout;
    }
    const int T = Tin, E = hc.embed, G = hc.conv_pos_groups, Eg = E / G, kp = hc.conv_pos;
    Ten* x = linear(layernorm(a, hp.ln), hp.proj.w, hp.proj.b, E, hp.proj.K);
    // padded frames -> 0 (wav2vec2.py:870-871); x += gelu(pos_conv(x)) with SamePad (:873-875)
    const int pad = kp / 2, Tp = T + kp;
    float* img = fm ? nullptr : alloc((long)G * B * Tp * Eg, true);
    bf16raw* imgh = fm ? alloc_h((long)G * B * Tp * Eg) : nullptr;
    if (fm && live()) hipMemsetAsync(imgh, 0, sizeof(bf16raw) * (size_t)G * B * Tp * Eg, st_);
    Ten* x2 = newT(B * T, E);
    if (live()) {
      chk(s2st_posconv_prep(x->d, frame_lens, img, imgh, B, T, E, G, pad, Tp, st_));
      // the G groups as ONE batched product (round 5: 16 launches of 150 tiles each -- a third of the CUs -- took 515 us of
      // the 5.8 ms forward): group z reads its image and its [Eg][kp * Eg] weights, writes columns [z Eg, (z + 1) Eg) of x2
      // (bias and residual follow the columns)
      constexpr bool each = false;
      for (int gi = 0; gi < 1; ++gi) {
        GemmArgs g{};
        const long io = (long)gi * B * Tp * Eg, wo = hp.pos_w + (long)gi * Eg * kp * Eg;
        g.A = fm ? gemm_rowmajor(imgh + io, Eg) : gemm_rowmajor(img + io, Eg);
        g.A.sp.per = T; g.A.sp.bs = (long)Tp * Eg;
        g.B = fm ? gemm_rowmajor(PH + wo, (long)kp * Eg) : gemm_rowmajor(P + wo, (long)kp * Eg);
        g.C = gemm_out(x2->d + (long)gi * Eg, E);
        g.ep = gemm_epi_default();
        g.ep.bias = P + hp.pos_b + (long)gi * Eg;
        g.ep.act = 2;
        g.ep.resid = x->d + (long)gi * Eg;
        g.M = B * T; g.N = Eg; g.K = kp * Eg; g.batch = 1; g.zdiv = 1; g.precise = c.precise;
        if (!each) {
          g.batch = G;
          g.A.zo = (long)B * Tp * Eg; g.B.zo = (long)Eg * kp * Eg; g.C.zo = Eg; g.ep.bias_zo = Eg;
        }
        chk(s2st_gemm(g, st_));
      }
    }
    Ten* y = layernorm(x2, hp.enc_ln);
    for (int l = 0; l < hc.layers; ++l) {
      const bool last = l == hc.layers - 1;
      const EncLayerP& L = hp.L[l];
      y = layernorm(self_attn_block(y, L.sa, B, T, hc.heads, frame_lens, 0, y), L.ln1);
      y = layernorm(ffn_block(y, L.fc1, L.fc2, y), L.ln2, last ? out : nullptr);
    }
    return err;
  }

  // ------------------------------------------------------------------------------------
  // S2ST_GEMM_STREAMK=1: bind stream-K scratch buffers to the two streams (opt-in: on the products of this step the
  // hand-off costs more than the idle tail it removes -- gemm_bf16.hip streamk_mode())
  bool use_streamk = s2st_env_on("S2ST_GEMM_STREAMK");
  void reset_call() {
    pending_wgrad.clear();
    pending_lnfold = s2st_lnfold_table{};
    s2st_gemm_streamk_unbind_all();  // the scratch lives in the previous call's workspace
    for (Ten* t : tens) delete t;
    tens.clear();
    tape.clear();
    tape_aware.clear();
    forked_ = false;
    in_region_ = false;
    marks.clear();
    ws_top = 0;
    ws_peak = 0;
    oom = false;
    err = 0;
    site = 0;
    site_log.clear();
    site_ctx[0] = 0;
    param_watermark = 0;
    next_segment = 0;
  }

  int forward() {
    const int B = bt.B, S = bt.S, D = bt.D, C = c.enc_dim, Cd = c.dec_dim;
    const int pad = c.conv_k / 2;
    const int T1 = c.text_input ? S : (S + 2 * pad - c.conv_k) / 2 + 1;
    const int T2 = c.text_input ? S : (T1 + 2 * pad - c.conv_k) / 2 + 1;
    if (T2 != bt.E) return S2ST_ERR_SHAPE;
    if (c.text_input && (!bt.src_txt || bt.Ls != S)) return S2ST_ERR_ARG;  // tokens are the encoder input
    const int E = T2;
    const bool tr = bt.training != 0;
    const bool with_loss = bt.tgt != nullptr;
    seed = bt.seed;
    if (tr) ensure_side();  // (also in the dry run that sizes the workspace: same stream set, same allocations)
    main_ = st_;
    forked_ = false;
    in_region_ = false;
    const bool two_chains = nchains == 2 && chain1_ && tr && fast();  // (training step only; see the note at chain_count)

    // sinusoidal tables come from the host side (cached per dim); conv weight layouts are
    // scratch at the bottom of the workspace
    const float *pe_enc = bt.pe_enc, *pe_dec = bt.pe_dec, *pe_asr = bt.pe_asr, *pe_st = bt.pe_st;
    const bool fm = fast();
    if (fm && !PH && !dry) return S2ST_ERR_ARG;
    // bf16 copy of the whole parameter arena (292 MB read + 146 MB written: ~0.08 ms)
    if (adam_pending && live() && (!fm || !ph_fresh)) adam_wait_all(st_);  // the whole arena is read right away
    if (fm && live()) {
      if (!ph_fresh) chk(s2st_cast_bf16_rows(P, n_params, PH, n_params, 1, (int)n_params, st_));
      ph_fresh = false;
    }
    // transposed weight copies for the backward, made on the second stream (idle during the forward)
    pht_valid = false;
    if (fm && tr && PHT && live()) {
      hipStream_t ts = side_ ? fork_side() : st_;
      // one launch per <= 200 matrices (the table rides in the kernel arguments)
      if (wt_tables.empty()) build_wt_tables();
      for (const s2st_transpose_table& tb : wt_tables) chk(s2st_transpose_bf16_batched(PH, PHT, tb, ts));
      pht_valid = true;
    }
    skws_n = fm ? (long)16 << 20 : 0;
    skws = fm ? alloc(skws_n) : nullptr;
    skws_side = fm && side_allowed ? alloc(skws_n) : skws;  // (by permission, not by existence: the stream is made lazily)
    skws_c1 = fm && chain1_ && nchains == 2 ? alloc(skws_n) : skws;
    // stream-K scratch of the persistent GEMM kernel, one per stream (ticket counters zeroed here, before any fork)
    if (fm && tr && use_streamk) {
      float* sk0 = alloc(S2ST_STREAMK_SCRATCH_FLOATS);
      float* sk1 = side_allowed ? alloc(S2ST_STREAMK_SCRATCH_FLOATS) : nullptr;
      if (live()) {
        hipMemsetAsync(sk0, 0, 4096, st_);
        s2st_gemm_streamk_bind(st_, sk0, S2ST_STREAMK_SCRATCH_FLOATS);
        if (sk1 && side_) {
          hipMemsetAsync(sk1, 0, 4096, st_);
          s2st_gemm_streamk_bind(side_, sk1, S2ST_STREAMK_SCRATCH_FLOATS);
        }
      }
    }
    typedef ConvW ConvScratch;
    auto conv_scratch = [&](const ConvP& p, bool need_wd) { return make_conv_scratch(p, need_wd, tr); };
    ConvScratch cs0{}, cs1{};
    std::vector<ConvScratch> cst;  // t2s encoder prenet
    if (c.text_input) {
      for (auto& pc : enc_conv) cst.push_back(conv_scratch(pc, true));
    } else {
      cs0 = conv_scratch(sub[0], false);
      cs1 = conv_scratch(sub[1], true);
    }
    std::vector<ConvScratch> csp;
    {
      // the post-net's weight layouts are first needed at the end of the forward: prepared on the second stream
      // (the data path meets that stream again at the first cross-attention, see cross_attn_block)
      const bool post_on_side = fm && side_ && ev_taps_ && ev_kv_ && live() && hoist_kv && !stop_after_encoder &&
                                c.dec_layers > 0;
      hipStream_t main_st = st_;
      if (post_on_side) st_ = fork_side();
      for (auto& pc : post_conv) csp.push_back(conv_scratch(pc, true));
      st_ = main_st;
    }

    mark();
    Ten* x = nullptr;
    if (c.text_input) {
      // ---- t2s text front (t2s_transformer.py:85-100): embedding -> conv/BatchNorm/ReLU prenet -> projection ->
      //      x += alpha * positions -> dropout ------------------------------------------------------------------
      Ten* emb = newT(B * E, C);
      touch(enc_embed + (long)c.src_vocab * C);
      if (live()) chk(s2st_embed_fwd((const long*)bt.src_txt, P + enc_embed, emb->d, B * E, C, 1.f, st_));
      const long eoff = enc_embed;
      tape.push_back([=]() {
        if (!emb->g) return;
        if (live()) chk(s2st_embed_bwd((const long*)bt.src_txt, emb->g, G + eoff, B * E, C, 1.f, 1, st_, ordered_sums ? c.src_vocab : 0));
      });
      set_ctx("enc.prenet");
      Ten* pn = text_prenet(emb, B, E, tr, cst);
      Ten* pj = linear(pn, enc_prenet_proj.w, enc_prenet_proj.b, C, C);
      set_ctx("enc.pe");
      x = add_pe(pj, bt.enc_pos, pe_enc, 1.f, enc_pos_alpha, tr ? c.dropout : 0.f);
    } else {
    // ---- encoder front: 2 x (conv k s2 -> GLU), sqrt(C) scale + positions + dropout -------------
    float* xh0 = fm ? nullptr : alloc((long)B * (S + 2 * pad) * c.in_dim, true);
    if (live() && !fm) {
      Split xs{(long)c.in_dim, 0, 0, 0};
      Split ys{(long)c.in_dim, (long)(S + 2 * pad) * c.in_dim, S, 0};
      chk(s2st_copy_rows(bt.src, xs, xh0 + (long)pad * c.in_dim, ys, B * S, c.in_dim, st_));
    }
    const bf16raw* xh0h = fm ? cast_halo(bt.src, B, S, pad, c.in_dim, true) : nullptr;
    Ten* z1 = conv(ConvIn{xh0, nullptr, S, xh0h}, sub[0], B, 2, cs0);
    const int C1 = c.conv_channels / 2;
    float* g1h = alloc((long)B * (T1 + 2 * pad) * C1, !fm);
    Ten* g1 = glu_to(z1, g1h + (long)pad * C1, Split{(long)C1, (long)(T1 + 2 * pad) * C1, T1, 0}, C1);
    const bf16raw* g1hh = fm ? cast_halo(g1h, B, T1, pad, C1) : nullptr;
    Ten* z2 = conv(ConvIn{g1h, g1, T1, g1hh}, sub[1], B, 2, cs1);
    float* x0d = alloc((long)B * E * C);
    Ten* x0 = glu_to(z2, x0d, Split{(long)C, 0, 0, 0}, C);
    set_ctx("enc.pe");
    x = add_pe(x0, bt.enc_pos, pe_enc, c.no_scale_embedding ? 1.f : sqrtf((float)C), -1,
                    tr ? c.dropout : 0.f, enc_spk, E);
    }
    mark();
    // ---- encoder layers, taps -----------------------------------------------------------------
    Ten *tap_asr = nullptr, *tap_st = nullptr;
    in_region_ = two_chains;  // ---- two utterance-half chains: the encoder layers + the final layer norm
    for (int i = 0; i < c.enc_layers; ++i) {
      set_ctx("enc.L%d", i);
      x = enc_layer(x, enc[i], B, E);
      if (i == c.tap_asr) tap_asr = x;
      if (i == c.tap_st) tap_st = x;
      if (i % 3 == 2) mark();
    }
    const bool t2s_spk = c.text_input && enc_spk >= 0 && bt.speaker != nullptr;
    Ten* enc_out = has_enc_ln ? layernorm(x, enc_ln, t2s_spk ? nullptr : outs.enc_out) : x;
    in_region_ = false;
    if (live()) sync_chains();
    if (t2s_spk) {
      // t2s_transformer.py:107-111: x = spk_emb_proj(cat[x, emb.expand(T)]) on EVERY position (padded ones included),
      // after the final layer norm.  The concatenation is materialised so that forward, data gradient and weight
      // gradient are the ordinary linear(); its backward splits the gradient into x's block and the table's rows.
      const int Sd = c.spk_dim;
      Ten* cat = newT(B * E, C + Sd);
      Ten* xin = enc_out;
      touch_spk(enc_spk + (long)c.n_speakers * Sd);
      if (live()) {
        chk(s2st_copy_rows(xin->d, Split{(long)C, 0, 0, 0}, cat->d, Split{(long)(C + Sd), 0, 0, 0}, B * E, C, st_));
        chk(s2st_speaker_fill_cols(spk_tab(enc_spk), (const long*)bt.speaker, cat->d, B, E, C + Sd, C, Sd, st_));
      }
      const long soff = enc_spk;
      tape.push_back([=]() {
        if (!cat->g) return;
        if (!c.spk_frozen && live())
          chk(s2st_speaker_cols_bwd(cat->g, (const long*)bt.speaker, B, E, C + Sd, C, Sd, c.n_speakers, G + soff, st_));
        if (xin->needs_grad) {
          bool acc;
          float* dx = gradbuf(xin, acc);
          if (live()) chk(s2st_split_cols(cat->g, C + Sd, dx, C, B * E, C, acc ? 1 : 0, st_));
        }
      });
      enc_out = linear(cat, enc_spk_proj.w, enc_spk_proj.b, C, C + Sd, 0, 0.f, nullptr, outs.enc_out);
    } else if (!has_enc_ln && outs.enc_out && live())  // post-LN encoder (t2s default): the last layer's output is the result
      hipMemcpyAsync(outs.enc_out, x->d, sizeof(float) * (size_t)x->n(), hipMemcpyDeviceToDevice, st_);
    // (mtl variant / CTC head without the aux ASR decoder: tap 0 is the RAW layer output -- no aux_asr_norm,
    // s2st_transformer_mtl.py:150-153 -- handed out as is for greedy CTC decoding, speech_generator_for_s2st_mtl.py:66-69)
    if (!c.has_asr && tap_asr && outs.tap0 && live())
      hipMemcpyAsync(outs.tap0, tap_asr->d, sizeof(float) * (size_t)tap_asr->n(), hipMemcpyDeviceToDevice, st_);
    if (c.has_asr && tap_asr) tap_asr = layernorm(tap_asr, asr_norm, outs.tap0);
    if (c.has_st && tap_st) tap_st = layernorm(tap_st, st_norm, outs.tap1);
    // the CTC head and the aux text decoders only need the encoder taps: they are issued (below, in tape
    // order) on the second stream behind this event and run next to the mel decoder
    // (t2s feature-level CTC head: it reads the DECODER's output, not an encoder tap -- nothing to overlap, and its
    // backward adds to feature_out's gradient like the post-net's: kept on the data-path stream)
    const bool aux_on_side = side_ && ev_taps_ && live() && overlap_aux && !(c.text_input && c.has_ctc);
    const bool kv_on_side = side_ && ev_taps_ && ev_kv_ && live() && hoist_kv && !stop_after_encoder;
    if (aux_on_side || kv_on_side) hipEventRecord(ev_taps_, st_);
    aux_wait_idx = tape.size();  // the tap layer-norm closures are the last ones pushed so far
    mark();
    if (stop_after_encoder) {  // decode_begin: the AR loop drives the decoder itself
      enc_out_keep = enc_out;
      return err;
    }
    if (c.s2t_mode) return forward_s2t(enc_out, with_loss);
    // The cross-attention K|V projections of every decoder layer only need the encoder output: they are
    // issued here, on the second stream, and run under the prenet and the first self-attention block (their
    // backward -- data gradients into the encoder output, weight gradients -- then runs after the layers').
    std::vector<Ten*> xkv(c.dec_layers, nullptr);
    if (hoist_kv) {
      hipStream_t main_st = st_;
      if (kv_on_side) {
        hipStreamWaitEvent(side_, ev_taps_, 0);
        st_ = side_;
        side_used = true;
      }
      for (int i = 0; i < c.dec_layers; ++i) xkv[i] = cross_kv(enc_out, dec[i].xa, Cd);
      if (kv_on_side) {
        hipEventRecord(ev_kv_, side_);
        st_ = main_st;
        kv_wait_ = true;
      }
      mark();
    }
    // ---- decoder: prenet (dropout always on), alpha * positions, layers ---------------------------
    Ten* prev = newT(B * D, c.out_dim, const_cast<float*>(bt.prev));
    prev->needs_grad = false;
    if (dec_spk >= 0 && bt.speaker) {
      // the speaker's row replaces the first input frame (s2st_transformer.py:441-444): a copy of prev_output_tokens
      // with row (b, 0) overwritten; its gradient there is the table's gradient
      Ten* pv = newT(B * D, c.out_dim);
      touch_spk(dec_spk + (long)c.n_speakers * c.out_dim);
      if (live()) {
        hipMemcpyAsync(pv->d, bt.prev, sizeof(float) * (size_t)pv->n(), hipMemcpyDeviceToDevice, st_);
        chk(s2st_speaker_set_rows(spk_tab(dec_spk), (const long*)bt.speaker, pv->d, B, D, c.out_dim, st_));
      }
      pv->needs_grad = tr && !c.spk_frozen;
      const long doff = dec_spk;
      tape.push_back([=]() {
        if (!pv->g || c.spk_frozen) return;
        if (live())
          chk(s2st_speaker_bwd(pv->g, (const long*)bt.speaker, B, D, 1, c.out_dim, c.n_speakers, 0.f, 0, G + doff, st_));
      });
      prev = pv;
    }
    Ten* h = prev;
    set_ctx("dec.prenet");
    for (int i = 0; i < c.prenet_layers; ++i)
      h = linear(h, prenet[i].w, prenet[i].b, prenet[i].N, prenet[i].K, 1, c.prenet_dropout);
    h = linear(h, prenet.back().w, prenet.back().b, Cd, c.prenet_dim);
    set_ctx("dec.pe");
    Ten* y = add_pe(h, bt.dec_pos, pe_dec, 1.f, pos_alpha, tr ? c.dropout : 0.f);
    mark();
    in_region_ = two_chains;  // ---- two chains again: decoder layers, final layer norm, the two output projections
    float* attn_out = nullptr;
    Ten* tap_dec_t = nullptr;
    for (int i = 0; i < c.dec_layers; ++i) {
      float* am = (i == c.dec_layers - 1 && bt.want_attn) ? outs.attn : nullptr;
      set_ctx("dec.L%d", i);
      y = dec_layer(y, enc_out, dec[i], B, D, E, c.dec_heads, c.dec_pre_ln != 0, bt.tgt_lens, am, xkv[i]);
      if (c.has_ctc_tgt && i == c.tap_dec) tap_dec_t = y;  // raw layer output (s2st_transformer_mtl.py:325-327)
      if (i % 2 == 1) mark();
    }
    (void)attn_out;
    if (has_dec_ln) y = layernorm(y, dec_ln);
    Ten* feat = linear(y, feat_proj.w, feat_proj.b, c.out_dim, Cd, 0, 0.f, nullptr, outs.feat);
    Ten* eos = linear(y, eos_proj.w, eos_proj.b, 1, Cd, 0, 0.f, nullptr, outs.eos);
    in_region_ = false;
    if (live()) sync_chains();
    set_ctx("post");
    Ten* post = postnet(feat, B, D, tr, csp, outs.post_feat);
    set_ctx("");
    // ---- mtl variant: CTC over the TARGET text on a decoder layer's output (s2st_loss_mtl.py:171-186: input lengths =
    //      decoder steps, targets = tgt_text incl. EOS) ------------------------------------------------------
    Ten* ctc_tgt_logits = nullptr;
    float *ctc_tgt_per = nullptr, *ctc_tgt_dl = nullptr;
    if (c.has_ctc_tgt && tap_dec_t) {
      ctc_tgt_logits = linear(tap_dec_t, ctc_proj_tgt.w, ctc_proj_tgt.b, c.tgt_vocab, Cd);
      if (with_loss) {
        ctc_tgt_per = alloc(B);
        float* lp = alloc((long)B * D * c.tgt_vocab);
        float* wsd = alloc(s2st_ctc_workspace_floats(B, D, bt.Lt));
        ctc_tgt_dl = tr ? alloc(ctc_tgt_logits->n()) : nullptr;
        if (live())
          chk(s2st_ctc(ctc_tgt_logits->d, (const long*)bt.tgt_txt, bt.Lt, bt.tgt_lens, bt.tgt_txt_lens, B, D, c.tgt_vocab,
                       lp, ctc_tgt_per, ctc_tgt_dl, ctc_tgt_dl ? c.ctc_tgt_weight / B : 0.f, wsd, st_));
      }
    }
    mark();
    // ---- CTC head on tap 0 (ctc_proj lives on the decoder, fed the encoder tap; :458-463) ----------
    hipStream_t main_st = st_;
    aux_lo_idx = tape.size();
    if (aux_on_side) {
      hipStreamWaitEvent(side_, ev_taps_, 0);
      st_ = side_;
      side_used = true;
    }
    Ten* ctc_logits = nullptr;
    if (c.has_ctc && tap_asr && !c.text_input) ctc_logits = linear(tap_asr, ctc_proj.w, ctc_proj.b, c.src_vocab, C);
    // t2s_transformer (criterions/t2s_loss.py:134-144): CTC of the SOURCE TEXT against the decoder's feature_out --
    // log_softmax(ctc_proj(feature_out)) [D, B, V], input lengths = decoder steps, targets = src_text, blank 0
    const bool t2s_ctc = c.text_input && c.has_ctc;
    if (t2s_ctc) ctc_logits = linear(feat, ctc_proj.w, ctc_proj.b, c.src_vocab, c.out_dim);
    const int ctc_T = t2s_ctc ? D : E;
    const int* ctc_ilens = t2s_ctc ? bt.tgt_lens : bt.ctc_in_lens;
    // The CTC sweep (one workgroup per utterance, ~E sequential steps: latency-bound, ~0.4 ms) runs on the
    // second stream next to the aux decoders and the other loss kernels; joined before the loss is finalised.
    float* ctc_per = (with_loss && c.has_ctc) ? alloc(B) : nullptr;
    float *ctc_lp = nullptr, *ctc_ws = nullptr, *ctc_dl = nullptr;
    if (with_loss && c.has_ctc && ctc_logits) {
      ctc_lp = (outs.ctc_lprobs && !t2s_ctc) ? outs.ctc_lprobs : alloc((long)B * ctc_T * c.src_vocab);
      ctc_ws = alloc(s2st_ctc_workspace_floats(B, ctc_T, bt.Ls));
      // training: the CTC gradient w.r.t. the logits comes out of the same alpha/beta sweep as the
      // loss, so it is produced here (per unit of upstream gradient) and only scaled in the backward
      ctc_dl = tr ? alloc(ctc_logits->n()) : nullptr;
      if (live()) {
        hipStream_t cs = aux_on_side ? st_ : fork_side();
        chk(s2st_ctc(ctc_logits->d, (const long*)bt.src_txt, bt.Ls, ctc_ilens, bt.src_txt_lens, B, ctc_T,
                     c.src_vocab, ctc_lp, ctc_per, ctc_dl, ctc_dl ? c.ctc_weight / B : 0.f, ctc_ws, cs));
      }
    }
    // ---- aux text decoders ---------------------------------------------------------------------------
    Ten *asr_logits = nullptr, *st_logits = nullptr;
    if (c.has_asr && tap_asr && bt.prev_src_txt)
      asr_logits = aux_decoder(asr, tap_asr, (const long*)bt.prev_src_txt, bt.src_txt_pos, bt.src_txt_lens, B,
                               bt.Ls, pe_asr, outs.asr_logits);
    if (c.has_st && tap_st && bt.prev_tgt_txt)
      st_logits = aux_decoder(st, tap_st, (const long*)bt.prev_tgt_txt, bt.tgt_txt_pos, bt.tgt_txt_lens, B,
                              bt.Lt, pe_st, outs.st_logits);
    st_ = main_st;
    aux_hi_idx = tape.size();
    aux_bwd_on_side = aux_on_side && tr;
    mark();
    // ---- losses (s2st_loss.py:219-257) -----------------------------------------------------------------
    if (with_loss) {
      float* stats = outs.stats;
      const float nr = (float)bt.ntokens, nf = nr * c.out_dim;
      float* loss_ws = alloc(3L * S2ST_LOSS_ORDERED_FLOATS);
      if (live()) {
        hipMemsetAsync(stats, 0, sizeof(float) * 32, st_);
        // ordered sums: the loss kernels leave per-workgroup sums, the finalize kernel adds them in workgroup order
        s2st_loss_parts lp{};
        float* ow[3] = {nullptr, nullptr, nullptr};
        if (ordered_sums)
          for (int q = 0; q < 3; ++q) lp.part[q] = ow[q] = loss_ws + (long)q * S2ST_LOSS_ORDERED_FLOATS;
        chk(s2st_mel_loss(feat->d, post->d, eos->d, bt.tgt, bt.tgt_lens, B, D, c.out_dim, c.bce_pos_weight,
                          stats + S2ST_STAT_L1_SUM, 0, 0, 0, nullptr, nullptr, nullptr, st_, ow[0], &lp.nblocks[0]));
        join_side();  // aux logits, CTC per-utterance losses
        if (asr_logits)
          chk(s2st_ls_ce(asr_logits->d, (const long*)bt.src_txt, B * bt.Ls, c.src_vocab, 1, c.label_smoothing,
                         stats + S2ST_STAT_ASR_NLL, nullptr, 0.f, st_, ow[1], &lp.nblocks[1]));
        if (st_logits)
          chk(s2st_ls_ce(st_logits->d, (const long*)bt.tgt_txt, B * bt.Lt, c.tgt_vocab, 1, c.label_smoothing,
                         stats + S2ST_STAT_ST_NLL, nullptr, 0.f, st_, ow[2], &lp.nblocks[2]));
        chk(s2st_loss_finalize(stats, ctc_per, B, nf, nr, c.w_l1, c.w_mse, c.w_eos, c.ctc_weight,
                               c.asr_weight, c.st_weight, c.label_smoothing, c.src_vocab, c.tgt_vocab,
                               (float)bt.src_txt_ntokens, (float)bt.tgt_txt_ntokens, st_, ctc_tgt_per, c.ctc_tgt_weight,
                               ordered_sums ? &lp : nullptr));
      }
      tape.push_back([=]() {
        // roots of the backward: d loss / d {feat, post, eos, logits}
        const float gs = gscale;
        bool a1, a2, a3;
        float* dfeat = gradbuf(feat, a1);
        float* dpost = gradbuf(post, a2);
        float* deos = gradbuf(eos, a3);
        if (live())
          chk(s2st_mel_loss(feat->d, post->d, eos->d, bt.tgt, bt.tgt_lens, B, D, c.out_dim, c.bce_pos_weight,
                            nullptr, gs * c.w_l1 / nf, gs * c.w_mse / nf, gs * c.w_eos / nr, dfeat, dpost, deos,
                            st_));
        if (ctc_logits) {
          bool a;
          float* dl = gradbuf(ctc_logits, a);
          if (live()) {
            if (ctc_dl) chk(s2st_dropout(ctc_dl, dl, ctc_logits->n(), gs, 0.f, 0, 0, st_));  // dl = gs * ctc_dl
            else
              chk(s2st_ctc(ctc_logits->d, (const long*)bt.src_txt, bt.Ls, ctc_ilens, bt.src_txt_lens, B, ctc_T,
                           c.src_vocab, ctc_lp, ctc_per, dl, gs * c.ctc_weight / B, ctc_ws, st_));
          }
        }
        if (ctc_tgt_logits && ctc_tgt_dl) {
          bool a;
          float* dl = gradbuf(ctc_tgt_logits, a);
          if (live()) chk(s2st_dropout(ctc_tgt_dl, dl, ctc_tgt_logits->n(), gs, 0.f, 0, 0, st_));  // dl = gs * d(ctc_tgt)
        }
        if (asr_logits) {
          bool a;
          float* dl = gradbuf(asr_logits, a);
          if (live())
            chk(s2st_ls_ce(asr_logits->d, (const long*)bt.src_txt, B * bt.Ls, c.src_vocab, 1, c.label_smoothing,
                           nullptr, dl, gs * c.asr_weight / (float)bt.src_txt_ntokens, st_));
        }
        if (st_logits) {
          bool a;
          float* dl = gradbuf(st_logits, a);
          if (live())
            chk(s2st_ls_ce(st_logits->d, (const long*)bt.tgt_txt, B * bt.Lt, c.tgt_vocab, 1, c.label_smoothing,
                           nullptr, dl, gs * c.st_weight / (float)bt.tgt_txt_ntokens, st_));
        }
      });
    }
    if (adam_pending && live()) adam_wait_all(st_);  // (parameters no op of this configuration reads)
    join_side();  // nothing of this forward is left running on the second stream when it returns in st_ order
    mark();
    return err;
  }

  // s2t_transformer_hubert + s2t_loss (s2t_transformer_me.py:308-330; criterions/s2t_loss.py:80-160): text decoder over the
  // encoder output, label-smoothed NLL summed over the non-pad tokens, accuracy counts.  The decoder's tokens ride in the
  // batch's source-text slots (the host chose them by --test-type); the dictionary is the TARGET one for both types
  // (s2t_transformer_me.py:268-283 builds embedding and output projection from task.target_dictionary).
  int forward_s2t(Ten* enc_out, bool with_loss) {
    const int B = bt.B;
    if (!bt.prev_src_txt || bt.Ls <= 0) return err;  // encoder only (forward_encoder)
    Ten* logits = aux_decoder(s2t, enc_out, (const long*)bt.prev_src_txt, bt.src_txt_pos, bt.src_txt_lens, B, bt.Ls,
                              bt.pe_asr, outs.asr_logits);
    mark();
    if (with_loss && bt.src_txt) {
      float* stats = outs.stats;
      float* loss_ws = alloc(3L * S2ST_LOSS_ORDERED_FLOATS);
      if (live()) {
        hipMemsetAsync(stats, 0, sizeof(float) * 32, st_);
        s2st_loss_parts lp{};
        float* ow = nullptr;
        if (ordered_sums) lp.part[1] = ow = loss_ws + S2ST_LOSS_ORDERED_FLOATS;
        chk(s2st_ls_ce(logits->d, (const long*)bt.src_txt, B * bt.Ls, c.tgt_vocab, 1, c.label_smoothing,
                       stats + S2ST_STAT_ASR_NLL, nullptr, 0.f, st_, ow, &lp.nblocks[1]));
        // (w_asr = 1 over "1 token": the SUM (1 - eps - eps_i) nll + eps_i smooth, eps_i = eps / (V - 1), s2t_loss.py:52-55)
        chk(s2st_loss_finalize(stats, nullptr, B, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, c.label_smoothing, c.tgt_vocab,
                               c.tgt_vocab, 1.f, 1.f, st_, nullptr, 0.f, ordered_sums ? &lp : nullptr));
      }
      tape.push_back([=]() {
        bool a;
        float* dl = gradbuf(logits, a);
        if (live())
          chk(s2st_ls_ce(logits->d, (const long*)bt.src_txt, B * bt.Ls, c.tgt_vocab, 1, c.label_smoothing, nullptr, dl,
                         gscale, st_));
      });
    }
    if (adam_pending && live()) adam_wait_all(st_);
    join_side();
    mark();
    return err;
  }

  int n_segments() const { return marks.empty() ? 0 : (int)marks.size() - 1; }

  // run tape closures of segment `seg` (0 = last part of the forward)
  int backward_segment(int seg) {
    int ns = n_segments();
    if (seg < 0 || seg >= ns) return S2ST_ERR_ARG;
    if (seg == 0) join_side();  // transposed weights (and anything else the forward left on the side stream)
    size_t hi = marks[ns - seg].tape_idx, lo = marks[ns - seg - 1].tape_idx;
    hipStream_t main_st = st_;
    main_ = st_;
    in_region_ = false;
    for (size_t i = hi; i-- > lo;) {
      if (aux_bwd_on_side && live()) {
        // the aux decoders' backward (CTC head + text decoders: many small kernels that only produce
        // the taps' gradients and parameter gradients) runs on the second stream next to the mel
        // decoder's backward; the data path waits for it right before the tap layer norms consume it
        if (i + 1 == aux_hi_idx && st_ == main_st) { flush_wgrad(); flush_lnfold(); st_ = fork_side(); }
        if (i + 1 == aux_lo_idx && st_ != main_st) { flush_wgrad(); flush_lnfold(); hipEventRecord(ev_auxb_, st_); st_ = main_st; }
        if (i + 1 == aux_wait_idx) wait_traced(main_st, ev_auxb_, "aux decoders' backward (tap gradients)");
      }
      // (a closure that does not launch per chain itself sees everything the second chain did)
      if (live() && !(i < tape_aware.size() && tape_aware[i])) sync_chains();
      tape[i]();
      if (err) break;
    }
    flush_wgrad();  // the segment's gradients are final once its launches are enqueued
    flush_lnfold();
    if (st_ != main_st) { hipEventRecord(ev_auxb_, st_); st_ = main_st; }
    if (live()) sync_chains();  // (the caller's stream is the one the next segment / the optimizer continues on)
    // The segment's weight gradients live on the second stream.  A caller that overlaps the gradient
    // all-reduce waits on that stream itself (s2st_engine_side_stream); the data path only joins once,
    // after the last segment, so it never stalls behind the weight-gradient backlog.
    if (seg == ns - 1) join_side();
    return err;
  }
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
