// engine_ops.h -- a fragment of struct s2st_engine (included INSIDE the struct body by engine.cpp; not a stand-alone
// header): the ops of the training graph: forward launch + backward closure (linear, layer norm, attention, layers, convolution, GLU, positions, aux text decoder).
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
