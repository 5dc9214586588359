// engine_runtime.h -- a fragment of struct s2st_engine (included INSIDE the struct body by engine.cpp; not a stand-alone
// header): per-call runtime: workspace arena, parameter touches / overlapped optimizer waits, the queues of weight-gradient products and partial-sum folds.
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
