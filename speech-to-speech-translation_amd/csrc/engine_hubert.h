// engine_hubert.h -- a fragment of struct s2st_engine (included INSIDE the struct body by engine.cpp; not a stand-alone
// header): the frozen HuBERT front end (config 4): parameters in GEMM-ready layouts and the forward.
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
