// engine_step.h -- a fragment of struct s2st_engine (included INSIDE the struct body by engine.cpp; not a stand-alone
// header): one step: the forward schedule (encoder, decoder, post-net, aux heads, losses) and the backward over tape segments.
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
