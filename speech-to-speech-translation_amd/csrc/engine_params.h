// engine_params.h -- a fragment of struct s2st_engine (included INSIDE the struct body by engine.cpp; not a stand-alone
// header): parameter construction: names, shapes and arena offsets of every tensor, forward-use order.
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
