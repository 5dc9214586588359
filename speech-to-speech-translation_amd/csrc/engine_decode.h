// engine_decode.h -- a fragment of struct s2st_engine (included INSIDE the struct body by engine.cpp; not a stand-alone
// header): incremental decoding: the mel decoder's AR step over key / value caches (config 5) and the aux text decoders' cached step (beam search).
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
