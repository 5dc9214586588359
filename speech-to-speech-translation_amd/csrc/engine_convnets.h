// engine_convnets.h -- a fragment of struct s2st_engine (included INSIDE the struct body by engine.cpp; not a stand-alone
// header): the convolutional sub-networks: Tacotron-2 post-net and the t2s text prenet (conv -> BatchNorm -> tanh / ReLU -> dropout).
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
