"""Size-independent properties of the hot path at BASELINE.json's full size (configs[1] recipe geometry,
a max-tokens 20000 batch of the synthetic Fisher corpus -- bench.py's workload), where the CPU oracle is
too slow to be the checker:

* a step is a pure function of (parameters, batch, seed): same seed -> bit-identical statistics,
  another seed -> different dropout masks;
* data-parallel sharding is exact where the model allows it: the summed statistics of the two halves of
  a batch equal those of the whole batch (evaluation mode: the post-net BatchNorm couples utterances in
  training mode, in the reference as well);
* gradient accumulation over micro-batches (update_freq, fairseq/trainer.py:760-800) is a sum;
* the bf16 fast path agrees with the bf16x3 precise path within the north-star tolerance (loss 1e-3);
* valid frames do not see batch padding: encoder output and pre-post-net features of an utterance are
  unchanged when the batch is padded further.
"""
import importlib

import numpy as np
import pytest
import torch

import s2st_oracle as O
from configs import CONFIGS
from test_engine import DATA, ENG, LOSS_KEYS
from synth_weights import synth_tensor

NO_DROP = dict(dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, prenet_dropout=0.0, postnet_dropout=0.0)


def _engine(backend, cfg, precise=False):
    eng = importlib.import_module(ENG)
    a = O.make_args(**cfg)
    e = eng.Engine(a, backend.device, precise=precise)
    for name, pv, gv, isb in e.named_views():
        pv.copy_(torch.from_numpy(synth_tensor(name, tuple(pv.shape), 0)))
    return a, e


@pytest.fixture(scope="module")
def workload():
    """The first two max-tokens batches of the bench corpus (longest utterances first)."""
    D = importlib.import_module(DATA)
    corpus = D.SyntheticFisherCorpus(n_utts=4096, seed=1234)
    batches = corpus.batches(max_tokens=20000, bsz_mult=8)
    order = np.random.RandomState(7).permutation(len(batches))
    return corpus, [batches[order[0]], batches[order[1]], list(batches[2][:8]) + list(batches[8][:8])]


def _need_gpu(backend):
    if backend.kind != "hip":
        pytest.skip("full-size properties run on the GPU")


def _rel(g, ref):
    return float((g - ref).norm()) / float(ref.norm())


def _grad_close(g, ref, tol):
    return _rel(g, ref) <= tol


def test_step_is_a_function_of_the_seed(backend, workload):
    """Same seed -> the same step: the forward sweep (every output and logged loss) is bit-identical -- all
    of its reductions, BatchNorm statistics included, are fixed-order (the loss sums too: per-workgroup sums added in
    workgroup order by the finalize kernel); so is every sum of the backward: the gradient arena is bit-identical."""
    _need_gpu(backend)
    corpus, b = workload
    a, e = _engine(backend, CONFIGS["base_recipe"])
    s = corpus.collate_batch(b[0])
    runs = []
    for seed in (11, 11, 12):
        o = e.forward(s, training=True, seed=seed)
        e.zero_grad()
        e.backward(1.0)
        backend.sync()
        runs.append(({k: o[k].clone() for k in ("feature_out", "eos_out", "encoder_out", "post_feat_out", "stats")},
                     e.grads.clone()))
    (o0, g0), (o1, g1), (o2, g2) = runs
    assert torch.isfinite(g0).all()
    for k in ("feature_out", "eos_out", "encoder_out", "post_feat_out"):
        assert torch.equal(o0[k], o1[k]), k
        assert not torch.equal(o0[k], o2[k]), k
    assert torch.equal(o0["stats"], o1["stats"]), (o0["stats"], o1["stats"])  # every logged sum and loss term
    # round 3: no sum of the backward uses atomics any more (bias / embedding / position-scale gradients, CTC occupancies,
    # attention bias partials, layer-norm parameter partials are all folded in a fixed order): the whole gradient arena
    # repeats BIT FOR BIT
    assert torch.equal(g1, g0), (_rel(g1, g0), int((g1 != g0).sum()))
    assert not _grad_close(g2, g0, 5e-2)


def test_two_utterance_half_chains_at_full_size(backend, workload, monkeypatch):
    """S2ST_CHAINS=2 on the benchmarked batch.  Dropout off: the same step as the one-chain schedule (forward outputs
    row by row from the same arithmetic, gradients up to the order of the partial-sum folds).  Recipe dropouts: chain 1
    draws its own masks, so the step is ANOTHER sample of the same random function -- it repeats bit for bit for a
    seed, and its logged losses sit where other seeds of the one-chain schedule sit."""
    _need_gpu(backend)
    corpus, b = workload
    s = corpus.collate_batch(b[0])

    def run(chains, cfg, seeds):
        monkeypatch.setenv("S2ST_CHAINS", str(chains))
        a, e = _engine(backend, cfg)
        out = []
        for seed in seeds:
            o = e.forward(s, training=True, seed=seed)
            e.zero_grad()
            e.backward(1.0)
            backend.sync()
            out.append((o["stats"].clone(), o["feature_out"].clone(), e.grads.clone()))
        del e
        return out

    nd = dict(CONFIGS["base_recipe"], **NO_DROP)
    (s1, f1, g1), = run(1, nd, [3])
    (s2, f2, g2), = run(2, nd, [3])
    assert torch.allclose(s1, s2, rtol=2e-6, atol=1e-6), (s1, s2)
    assert float((f1 - f2).abs().max()) <= 1e-5 * float(f1.abs().max())
    assert _rel(g2, g1) <= 2e-4, _rel(g2, g1)  # (bf16 operands of the backward emitted from differently ordered fp32 sums)
    one = run(1, CONFIGS["base_recipe"], [11, 12, 13, 14])
    two = run(2, CONFIGS["base_recipe"], [11, 11, 12])
    assert torch.equal(two[0][0], two[1][0]) and torch.equal(two[0][2], two[1][2])
    assert not torch.equal(two[0][2], one[0][2])
    l1 = torch.stack([r[0][0] for r in one]).double()
    l2 = torch.stack([two[0][0][0], two[2][0][0]]).double()
    spread = float(l1.max() - l1.min())
    assert float((l2 - l1.mean()).abs().max()) <= max(3.0 * spread, 2e-2 * float(l1.mean())), (l1, l2)
    assert abs(_rel(two[0][2], one[0][2]) - _rel(one[1][2], one[0][2])) <= 0.5 * _rel(one[1][2], one[0][2])


# normalisation of each logged loss (s2st_loss.py:179-292, reduction="mean"): masked frames for the
# spectrogram / stop losses, utterances for CTC (nn.CTCLoss mean), text tokens for the aux decoders
LOSS_WEIGHT = {"l1_loss": "ntokens", "mse_loss": "ntokens", "eos_loss": "ntokens", "ctc_loss": "nsentences",
               "aux_asr_loss": "src_txt_ntokens", "aux_st_loss": "tgt_txt_ntokens"}


def test_shards_of_a_batch_add_up(backend, workload):
    """What data parallelism relies on: every logged loss of a batch is the weighted mean of its shards'
    (evaluation mode: the post-net BatchNorm couples utterances in training mode, as in the reference)."""
    _need_gpu(backend)
    corpus, b = workload
    a, e = _engine(backend, dict(CONFIGS["base_recipe"], **NO_DROP))
    ix = list(b[0])
    sw = corpus.collate_batch(ix)
    whole = e.forward(sw, training=False, seed=1)["stats"].double().cpu()
    parts = []
    for h in (ix[0::2], ix[1::2]):
        sh = corpus.collate_batch(h)
        parts.append((sh, e.forward(sh, training=False, seed=1)["stats"].double().cpu()))
    for k, i in LOSS_KEYS:
        if k == "loss":
            continue
        w = LOSS_WEIGHT[k]
        got = sum(float(st[i]) * sh[w] for sh, st in parts) / sw[w]
        assert abs(got - float(whole[i])) <= 2e-4 * max(1.0, abs(float(whole[i]))), (k, got, float(whole[i]))


def test_gradient_accumulation_is_a_sum(backend, workload):
    _need_gpu(backend)
    corpus, b = workload
    a, e = _engine(backend, CONFIGS["base_recipe"], precise=True)
    sa, sb = corpus.collate_batch(b[0]), corpus.collate_batch(b[1])
    single = []
    for s, seed in ((sa, 5), (sb, 6)):
        e.forward(s, training=True, seed=seed)
        e.zero_grad()
        e.backward(1.0)
        backend.sync()
        single.append(e.grads.clone())
    e.zero_grad()
    for s, seed in ((sa, 5), (sb, 6)):
        e.forward(s, training=True, seed=seed)
        e.backward(1.0)
    backend.sync()
    assert _rel(e.grads, single[0] + single[1]) <= 5e-5, _rel(e.grads, single[0] + single[1])


def test_fast_path_within_north_star_tolerance_of_precise(backend, workload):
    _need_gpu(backend)
    corpus, b = workload
    s = corpus.collate_batch(b[0])
    out = []
    for precise in (True, False):
        a, e = _engine(backend, dict(CONFIGS["base_recipe"], **NO_DROP), precise=precise)
        o = e.forward(s, training=True, seed=1)
        e.zero_grad()
        e.backward(1.0)
        backend.sync()
        out.append((o["stats"].double().cpu(), e.grads.clone()))
        del e
    (sp, gp), (sf, gf) = out
    tot = dict(LOSS_KEYS)["loss"]
    assert abs(float(sf[tot]) - float(sp[tot])) <= 1e-3 * abs(float(sp[tot])), (float(sf[tot]), float(sp[tot]))
    for k, i in LOSS_KEYS:  # components: within 5e-3 of themselves or 1e-3 of the total
        assert abs(float(sf[i]) - float(sp[i])) <= max(5e-3 * abs(float(sp[i])), 1e-3 * abs(float(sp[tot]))), k
    assert abs(float(gf.norm()) - float(gp.norm())) <= 1e-2 * float(gp.norm())
    assert _grad_close(gf, gp, 5e-2)


class _SampledGrads:
    """The oracle's gradients in the layout of a golden file's ``gsub.*`` entries (test_engine.check_gradient_direction)."""
    def __init__(self, named):
        from test_engine import gsub
        self.d = {"gsub." + n: gsub(g.detach().numpy()) for n, g in named.items()}
        self.files = list(self.d)

    def __getitem__(self, k):
        return self.d[k]


# Which max-tokens 20000 batches of the bench corpus meet the oracle (index into bench.py's shuffled order; geometry =
# utterances x longest source frames -> encoder length E, decoder steps D).  bench.py's 20 timed batches are order[W ... W + 19]:
#   0: 16 x 850  (E 213, D 140)  -- several 32-row query blocks per key block, two 128-key blocks in the encoder
#   1: 40 x 429  (E 108, D 72)   -- the median geometry: one key block
#  16: 8 x 1394  (E 349, D 219)  -- the longest timed batch: three key blocks, causal tile skipping over 7 query blocks
#   8: 184 x 107 (E 27, D 18)    -- many short utterances: one partial tile per (batch, head) pair, 736 pairs
FULL_BATCHES = {"16x850": 0, "40x429": 1, "8x1394": 16, "184x107": 8}


@pytest.fixture(scope="module", params=list(FULL_BATCHES), ids=list(FULL_BATCHES))
def oracle_full_batch(request, workload):
    """The CPU oracle (torch fp32) on ALL utterances of a max-tokens 20000 batch of the bench corpus, dropouts 0, training
    mode (BatchNorm batch statistics): forward, criterion, backward -- ~3 - 6 s on the GPU host's cores, once per batch
    (computed at first use: the CPU suite, which skips the test, does not pay for it)."""
    from test_engine import make_oracle
    corpus, b = workload
    cache = []

    def get():
        if not cache:
            batches = corpus.batches(max_tokens=20000, bsz_mult=8)
            order = np.random.RandomState(7).permutation(len(batches))
            s = corpus.collate_batch(batches[order[FULL_BATCHES[request.param]]])
            _, m = make_oracle(dict(CONFIGS["base_recipe"], **NO_DROP))
            loss, ss, log, outs = O.criterion_forward(m, s)
            loss.backward()
            grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
            outs = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in outs.items()}
            cache.append((s, log, outs, grads))
        return cache[0]
    return get


@pytest.mark.parametrize("precise", [True, False], ids=["bf16x3", "bf16"])
def test_full_batch_against_oracle(backend, workload, oracle_full_batch, golden_dir, precise):
    """at the BENCHMARKED batch sizes (FULL_BATCHES: the 16 x 850-frame batch, the median
    40 x 429 one, the longest timed batch 8 x 1394 and the 184-utterance one) the tile
    picker's 4-wave forms, the one-round 128 x 128 rule, the grouped weight-gradient launch with one XCD run per tile range
    and the XCD-aware attention block order engage -- and were only ever compared with the precise mode of the same
    library.  Here the whole path is held against the ORACLE on the bench's first batch: every loss term (bf16x3 5e-5,
    bf16 1e-3 = north-star), sampled outputs (flat[::127], not checksums), the direction of every gradient tensor (bf16x3:
    5e-3 per tensor / 2e-3 whole; bf16: the bounds derived from the reference under autocast, test_engine.bf16_tensor_bounds),
    and in bf16x3 mode the integer outputs bit for bit (stop indices, greedy CTC path, encoder lengths)."""
    _need_gpu(backend)
    from test_engine import bf16_tensor_bounds, check_gradient_direction, gsub
    s, log, outs, ograds = oracle_full_batch()
    a, e = _engine(backend, dict(CONFIGS["base_recipe"], **NO_DROP), precise=precise)
    o = e.forward(s, training=True, want_attn=True, seed=1)
    e.zero_grad()
    e.backward(1.0)
    backend.sync()
    st = o["stats"].double().cpu()
    ltol = 5e-5 if precise else 1e-3
    tot = float(log["loss"])
    for k, i in LOSS_KEYS:
        ref = float(log[k])
        if precise or k == "loss":  # north-star: loss parity 1e-3
            assert abs(float(st[i]) - ref) < ltol * max(1.0, abs(ref)), (k, float(st[i]), ref)
        else:  # bf16 components: 5e-3 of themselves or 1e-3 of the total (test_fast_path_within_north_star_tolerance_of_precise)
            assert abs(float(st[i]) - ref) <= max(5e-3 * abs(ref), 1e-3 * abs(tot)), (k, float(st[i]), ref)
    assert int(st[5]) == log["asr_n_correct"] or not precise
    otol = 5e-4 if precise else 3e-2
    pairs = [("encoder_out", outs["encoder_out"].transpose(0, 1)), ("feature_out", outs["feature_out"]),
             ("eos_out", outs["eos_out"]), ("post_feat_out", outs["post_feat_out"]), ("attn", outs["attn"]),
             ("asr_logits", outs["asr_logits"]), ("st_logits", outs["st_logits"]),
             ("ctc_lprobs", outs["ctc_lprobs"].transpose(0, 1))]
    for k, ref in pairs:
        mine = gsub(o[k].detach().cpu().contiguous().numpy())
        r = gsub(ref.contiguous().numpy())
        assert mine.shape == r.shape, (k, mine.shape, r.shape)
        if k == "ctc_lprobs":  # (log-probabilities: padded frames hold -inf-like fill on neither side, compare finite ones)
            ok = np.isfinite(r) & np.isfinite(mine)
            mine, r = mine[ok], r[ok]
        err = float(np.abs(mine - r).max()) / float(np.abs(r).max())
        assert err < otol, (k, err)
    assert torch.equal(o["encoder_lens"].cpu().long(), outs["encoder_lens"])
    if precise:
        assert torch.equal(O.stop_indices(o["eos_out"].cpu()), O.stop_indices(outs["eos_out"]))
        il = O.ctc_input_lengths(s["net_input"]["src_speech_lens"], [5, 5])
        assert torch.equal(O.ctc_greedy_path(o["ctc_lprobs"].cpu().transpose(0, 1), il), O.ctc_greedy_path(outs["ctc_lprobs"], il))
    grads = {n: gv for n, pv, gv, isb in e.named_views() if not isb}
    z = _SampledGrads(ograds)
    tol_of, whole_tol, ac = bf16_tensor_bounds(golden_dir)
    w, whole = check_gradient_direction(grads, z, 5e-3 if precise else tol_of, 2e-3 if precise else whole_tol,
                                        tag="full batch " + ("bf16x3" if precise else "bf16"))
    print(f"[full batch vs oracle {'bf16x3' if precise else 'bf16'}] worst tensor {w[1]} {w[0]:.2e}, whole gradient {whole:.2e}")


@pytest.mark.parametrize("precise,tol", [(True, 1e-4), (False, 2e-2)])
def test_valid_frames_do_not_see_batch_padding(backend, workload, precise, tol):
    """(bf16 mode: a different padded length changes tile counts, hence fp32 summation order in the last bit,
    which operand rounding amplifies -- the tolerance is a few bf16 ulps of the largest activation.)"""
    _need_gpu(backend)
    corpus, b = workload
    a, e = _engine(backend, dict(CONFIGS["base_recipe"], **NO_DROP), precise=precise)
    ix = list(b[2])  # two length buckets in one batch
    s = corpus.collate_batch(ix)
    o = e.forward(s, training=False, seed=1)
    enc, feat = o["encoder_out"].clone(), o["feature_out"].clone()  # [B, T', C], [B, D, 80 * r]
    enc_len = o["encoder_lens"].tolist()
    s2 = corpus.collate_batch(ix)
    ni = s2["net_input"]
    ni["src_speech"] = torch.nn.functional.pad(ni["src_speech"], (0, 0, 0, 64))
    ni["prev_output_tokens"] = torch.nn.functional.pad(ni["prev_output_tokens"], (0, 0, 0, 5))
    s2["tgt_speech"] = torch.nn.functional.pad(s2["tgt_speech"], (0, 0, 0, 5))
    o2 = e.forward(s2, training=False, seed=1)
    backend.sync()
    tl = s["target_lengths"].tolist()
    sl = ni["src_speech_lens"].tolist()
    # the reference's conv subsampler does not mask between its layers, so an utterance that ends within the
    # receptive field (2 x k5 s2 -> 7 frames) of the batch edge sees the difference between the conv's zero
    # padding and what the first layer makes of batch padding; those are excluded, as they would be there
    inner = [bi for bi in range(len(ix)) if sl[bi] + 8 <= max(sl)]
    assert len(inner) >= 2
    for bi in inner:
        d = tl[bi]
        assert float((o2["feature_out"][bi, :d] - feat[bi, :d]).abs().max()) <= tol * float(feat[bi, :d].abs().max())
        t = enc_len[bi]
        ref = enc[bi, :t]
        assert float((o2["encoder_out"][bi, :t] - ref).abs().max()) <= tol * float(ref.abs().max())


@pytest.mark.parametrize("precise", [False, True])
def test_longest_admissible_utterances(backend, precise):
    """The size filter admits sources up to --max-source-positions = 3000 frames (speech_to_text_dataset.py:346-347):
    a max-tokens batch of such utterances (6 x 3000, 750 encoder frames, ~450 decoder steps) must run on both GEMM
    paths, and a second, short batch right after it must be unaffected by the larger one's leftovers in the
    workspace (same result as on a fresh engine)."""
    _need_gpu(backend)
    D = importlib.import_module(DATA)
    long_c = D.SyntheticFisherCorpus(n_utts=6, seed=2, min_src=2990, max_src=3000, median_src=3000.0, sigma=0.001)
    assert int(long_c.src_n_frames.max()) == 3000
    short_c = D.SyntheticFisherCorpus(n_utts=8, seed=4, min_src=40, max_src=60, median_src=50.0)
    a, e = _engine(backend, dict(CONFIGS["base_recipe"], **NO_DROP), precise=precise)
    sl, ss = long_c.collate_batch(range(6)), short_c.collate_batch(range(8))
    o = e.forward(sl, training=True, seed=1)
    e.zero_grad()
    e.backward(1.0)
    backend.sync()
    assert torch.isfinite(o["stats"]).all() and torch.isfinite(e.grads).all()
    assert o["encoder_out"].shape[1] == 750
    o2 = e.forward(ss, training=True, seed=1)
    e.zero_grad()
    e.backward(1.0)
    backend.sync()
    st2, g2, outs2 = o2["stats"].clone(), e.grads.clone(), {k: o2[k].clone() for k in ("feature_out", "post_feat_out", "eos_out")}
    del e
    a, f = _engine(backend, dict(CONFIGS["base_recipe"], **NO_DROP), precise=precise)
    o3 = f.forward(ss, training=True, seed=1)
    f.zero_grad()
    f.backward(1.0)
    backend.sync()
    for k, v in outs2.items():
        assert torch.equal(o3[k], v), k
    assert torch.allclose(o3["stats"], st2, rtol=1e-6, atol=0)  # (loss sums use atomics: last-bit order noise)
    # (the fp32-operand GEMM of the precise path splits K with atomics: order noise of a few 1e-6)
    assert _rel(f.grads, g2) <= (2e-5 if precise else 1e-6), _rel(f.grads, g2)


def test_overlapped_optimizer_update_gives_the_same_trajectory(backend, workload):
    """``train_step(overlap_optimizer=True)``: the Adam kernel runs in chunks on the engine's second stream and the next
    forward waits chunk by chunk (s2st_engine_adam_overlapped).  Every sum of a step is ordered, so the trajectory is a
    bit-exact function of (parameters, batches, seeds): five updates at base size with the recipe's dropouts end on the
    SAME bits with and without the overlap -- a forward that read a parameter chunk before its update landed would not.
    (On the emulator there is no second stream: the chunked update itself is what is compared.)"""
    PKG = ENG.rsplit(".runtime", 1)[0]
    tasks = importlib.import_module(PKG + ".tasks")
    tr = importlib.import_module(PKG + ".trainer")
    from synth_weights import load_synth
    big = backend.kind == "hip"
    corpus, b = workload
    if big:
        cfg = CONFIGS["base_recipe"]
        batches = [corpus.collate_batch(x) for x in b]
    else:
        from test_engine import NANO, nano_batches
        cfg = dict(NANO, dropout=0.1, attention_dropout=0.1)
        batches = nano_batches()
    runs = []
    for overlap in (False, True, True):
        a = O.make_args(**cfg)
        a.precise_gemm, a.lr, a.warmup_updates, a.clip_norm = False, 1e-3, 2, 0.05
        task = tasks.S2ST_TranslationTask.setup_task(a, device=backend.device)
        model = task.build_model(a)
        load_synth(model, 0)
        t = tr.Trainer(a, task, model, task.build_criterion(a))
        gn = []
        for u in range(5):
            r = t.train_step([batches[u % len(batches)]], overlap_optimizer=overlap)
            gn.append(r["gnorm"])  # (read after the loop: no host sync between the updates)
        t.wait_optimizer()
        backend.sync()
        runs.append((model.engine.params.clone(), t.exp_avg.clone(), float(gn[-1][0]), int(t.skipped)))
        del t, model, task
    (p0, m0, g0, s0), (p1, m1, g1, s1), (p2, m2, g2, s2) = runs
    assert s0 == s1 == s2 == 0 and np.isfinite(g0)
    assert torch.equal(p1, p2) and torch.equal(m1, m2)  # the overlapped schedule repeats itself ...
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and g0 == g1  # ... and equals the plain one
