"""Parity with the recipe's dropouts ON.

The reference draws a fresh mask from torch's generator at every ``F.dropout`` (sites: transformer_layer.py:150-162,
384-431; multihead_attention.py:360-366 on the probabilities; s2st_transformer.py:197-208, 385-388 after the positions;
tacotron2.py:95-98 Prenet, ALWAYS on; tacotron2.py:122-126 Postnet after every conv incl. the last;
transformer_decoder.py:281-370 for the aux heads).  The HIP engine keeps no masks: keep(element) = hash(site seed,
element index), evaluated inside the fused epilogues / attention tiles and AGAIN in the backward.  So the step is run on
the engine first with its site log on (``s2st_engine_site_log``), the masks are regenerated per site through the C ABI
(``s2st_dropout_f32`` over ones: the same hash) and INJECTED into the CPU oracle (``oracle.injected_masks``), whose
placement, 1 / (1 - p) scaling and autograd are the reference's.  Then everything is compared as in the dropout-off tests:
every loss term, outputs, the direction of every gradient tensor, BatchNorm running statistics.  A site applied before
instead of after a residual add, a missing scale on one epilogue, a forward / backward mask mismatch or a wrong element
index in a fused kernel fails here; the test also demands that the two sides have exactly the same SET of sites.
"""
import importlib
import math

import pytest
import torch

import s2st_oracle as O
from configs import CONFIGS, golden_sample
from test_engine import DATA, LOSS_KEYS, MICRO, MICRO_POSTLN, make_engine, make_oracle, rel

RECIPE = dict(dropout=0.1, attention_dropout=0.1, activation_dropout=0.01, prenet_dropout=0.5, postnet_dropout=0.5)
# (the micro models are 64 wide: activation dropout 0.01 would drop ~1 element per row -- use a value that bites)
STRONG = dict(dropout=0.1, attention_dropout=0.1, activation_dropout=0.05, prenet_dropout=0.5, postnet_dropout=0.5)
WIDE = dict(encoder_embed_dim=128, decoder_embed_dim=128, encoder_attention_heads=2, decoder_attention_heads=2)


def provider_from(e):
    """oracle.drop() provider over the engine's last forward: site name -> keep mask in the oracle's layout."""
    sites = e.dropout_sites()
    used = {}

    def provider(site, layout, x, p):
        assert site in sites, (site, "the engine has no such dropout site", sorted(sites))
        r = sites[site]
        assert abs(r.p - p) < 1e-7, (site, r.p, p)
        assert site not in used, (site, "asked twice")
        keep = e.dropout_keep_mask(r).cpu()
        used[site] = float(keep.mean())
        if layout == "attn":  # engine [B][H][T][ld] -> oracle [B*H, T, S]
            B, H, T, ld = keep.shape
            S = x.shape[-1]
            assert int(r.dims[3]) == S and x.shape[0] == B * H and x.shape[1] == T, (site, tuple(x.shape), list(r.dims))
            return keep[..., :S].reshape(B * H, T, S)
        rows, N = keep.shape
        if layout == "btc":
            B = x.shape[0]
            return keep.view(B, rows // B, N)
        if layout == "tbc":
            B = x.shape[1]
            return keep.view(B, rows // B, N).transpose(0, 1)
        assert layout == "bct", layout
        B = x.shape[0]
        return keep.view(B, rows // B, N).permute(0, 2, 1)

    return provider, used, sites


def check_with_injected_masks(backend, e, m, sample, out_tol, loss_tol, grad_tol, whole_tol, seed=11, stat_tol=2e-4):
    """Engine step (site log on) -> its masks into the oracle -> compare.  Returns (engine outputs, oracle log)."""
    O.name_sites(m)
    e.site_log(True)
    o = e.forward(sample, training=True, want_attn=True, seed=seed)
    e.zero_grad()
    e.backward(1.0)
    backend.sync()
    provider, used, sites = provider_from(e)
    with O.injected_masks(provider):
        loss, ss, log, outs = O.criterion_forward(m, sample)
    loss.backward()
    # the same SET of sites on both sides, each with a plausible keep rate
    assert set(used) == set(sites), (sorted(set(sites) - set(used)), sorted(set(used) - set(sites)))
    for name, rate in used.items():
        n = 1
        for d in sites[name].dims[:2]:
            n *= int(d)
        assert abs(rate - (1.0 - sites[name].p)) < 6.0 / math.sqrt(max(n, 1)) + 0.02, (name, rate, sites[name].p)
    pairs = [("encoder_out", outs["encoder_out"].transpose(0, 1)), ("feature_out", outs["feature_out"]),
             ("eos_out", outs["eos_out"]), ("post_feat_out", outs["post_feat_out"]), ("attn", outs["attn"])]
    if outs["asr_logits"] is not None:
        pairs += [("asr_logits", outs["asr_logits"]), ("st_logits", outs["st_logits"])]
    if outs["ctc_lprobs"] is not None:
        pairs += [("ctc_lprobs", outs["ctc_lprobs"].transpose(0, 1))]
    for k, ref in pairs:
        assert rel(o[k], ref) < out_tol, (k, rel(o[k], ref))
    st = o["stats"].cpu()
    for k, i in LOSS_KEYS:
        assert abs(float(st[i]) - float(log[k])) < loss_tol * max(1.0, abs(float(log[k]))), (k, float(st[i]), float(log[k]))
    if "asr_n_correct" in log and out_tol < 1e-3:
        assert int(st[5]) == log["asr_n_correct"] and int(st[6]) == log["asr_total"]
        assert int(st[9]) == log["st_n_correct"] and int(st[10]) == log["st_total"]
    if out_tol < 1e-3:
        assert torch.equal(O.stop_indices(o["eos_out"].cpu()), O.stop_indices(outs["eos_out"]))
    # gradient direction of every tensor + the whole gradient
    named = dict(m.named_parameters())
    gmax = max(float(p.grad.norm()) for p in named.values() if p.grad is not None)
    err2 = ref2 = 0.0
    worst = []
    for name, pv, gv, isb in e.named_views():
        if isb:
            continue
        rg = named[name].grad
        rg = torch.zeros_like(named[name]) if rg is None else rg
        d = (gv.cpu() - rg).double()
        worst.append((float(d.norm()) / (float(rg.double().norm()) + 1e-3 * gmax), name))
        err2 += float(d.norm()) ** 2
        ref2 += float(rg.double().norm()) ** 2
    worst.sort(reverse=True)
    whole = math.sqrt(err2 / ref2)
    print(f"[dropout parity] {len(sites)} sites; whole gradient {whole:.2e}; worst " +
          ", ".join(f"{n} {v:.2e}" for v, n in worst[:4]))
    assert worst[0][0] < grad_tol, worst[:5]
    assert whole < whole_tol, whole
    # BatchNorm running statistics after this one training forward (dropout sits BEHIND each BatchNorm: a mask applied in
    # front of the statistics would show here)
    bufs = dict(m.named_buffers())
    for name, pv, gv, isb in e.named_views():
        if isb and name in bufs and not name.endswith("num_batches_tracked"):
            ref = bufs[name]
            assert float((pv.cpu() - ref).abs().max()) < stat_tol * (float(ref.abs().max()) + 1e-2), name
    return o, log


def micro_batch():
    D = importlib.import_module(DATA)
    c = D.SyntheticFisherCorpus(n_utts=4, seed=3, max_src=64, median_src=50, min_src=30)
    return c.collate_batch(range(4))


@pytest.mark.parametrize("cfg", [MICRO, MICRO_POSTLN], ids=["preln_aux", "postln"])
def test_micro_recipe_dropouts_against_oracle_precise(backend, cfg):
    """bf16x3 mode (unfused attention, fp32-accurate products): every site's placement and scaling, forward and backward."""
    cfg = dict(cfg, **STRONG)
    a, e = make_engine(backend, cfg, precise=True)
    _, m = make_oracle(cfg)
    check_with_injected_masks(backend, e, m, micro_batch(), out_tol=3e-4, loss_tol=3e-5, grad_tol=5e-3, whole_tol=2e-3)


@pytest.mark.parametrize("cfg", [MICRO, MICRO_POSTLN], ids=["preln_aux", "postln"])
def test_micro_recipe_dropouts_against_oracle_fast(backend, cfg):
    """The benchmarked bf16 mode at head width 64: fused attention kernels (masks regenerated per score tile in the forward
    AND in both backward passes), dropout in the GEMM epilogues, the layer-norm backward's fused dropout-backward operand,
    the ReLU-dropout backward in the data-gradient epilogue, BatchNorm + tanh + dropout written as the next conv's image."""
    if backend.kind == "emu" and cfg is MICRO_POSTLN:
        pytest.skip("post-LN in fast mode: covered by the GPU run (the CPU suite's time budget)")
    cfg = dict(cfg, **STRONG, **WIDE)
    a, e = make_engine(backend, cfg, precise=False)
    _, m = make_oracle(cfg)
    check_with_injected_masks(backend, e, m, micro_batch(), out_tol=3e-2, loss_tol=2e-3, grad_tol=0.5, whole_tol=0.12,
                              stat_tol=4e-2)


def test_micro_text_input_variant_recipe_dropouts_against_oracle(backend):
    """t2s_transformer (text encoder: conv -> BatchNorm -> ReLU -> dropout prenet, t2s_transformer.py:55-66, 86-100; post-LN
    encoder layers): the BatchNorm + ReLU + dropout sites of the encoder prenet (``enc.prenet/norm<i>``), the dropout behind
    the alpha-scaled positions and every layer site, bf16x3 mode -- the same mask-injection comparison."""
    cfg = dict(MICRO, asr_ce_weight=0.0, st_ce_weight=0.0, ctc_weight=0.0, text_encoder=True, encoder_conv_layers=2,
               encoder_conv_kernel_size=5, encoder_dropout=0.3, encoder_normalize_before=False, **STRONG)
    a, e = make_engine(backend, cfg, precise=True)
    _, m = make_oracle(cfg)
    o, log = check_with_injected_masks(backend, e, m, micro_batch(), out_tol=3e-4, loss_tol=3e-5, grad_tol=1e-2, whole_tol=2e-3)
    assert any(k.startswith("enc.prenet/norm") for k in e.dropout_sites())


def test_masks_are_what_the_comparison_rests_on(backend):
    """Control: the SAME comparison with the masks of a different seed must fail loudly (losses move by >> the tolerance):
    the test above is not passing because dropout is too weak to matter."""
    cfg = dict(MICRO, **STRONG)
    a, e = make_engine(backend, cfg, precise=True)
    _, m = make_oracle(cfg)
    O.name_sites(m)
    s = micro_batch()
    e.site_log(True)
    o = e.forward(s, training=True, want_attn=True, seed=11)
    backend.sync()
    mine = float(o["stats"][16])
    e.forward(s, training=True, want_attn=True, seed=12)  # the log now holds seed 12's sites
    backend.sync()
    provider, used, sites = provider_from(e)
    with O.injected_masks(provider):
        loss, ss, log, outs = O.criterion_forward(m, s)
    assert abs(mine - float(log["loss"])) > 1e-2 * abs(mine), (mine, float(log["loss"]))


@pytest.mark.gpu
@pytest.mark.parametrize("precise", [True, False], ids=["bf16x3", "bf16"])
@pytest.mark.parametrize("name", ["tiny", "tiny_postln"])
def test_tiny_recipe_dropouts_against_oracle(backend, name, precise):
    """BASELINE.json configs[0] geometry (d 128, 4 heads: head width 32 -> unfused attention in both modes) with the
    recipe's dropout values on the golden batch."""
    if backend.kind != "hip":
        pytest.skip("golden-size configs run on the GPU")
    cfg = dict(CONFIGS[name], **RECIPE)
    a, e = make_engine(backend, cfg, precise=precise)
    _, m = make_oracle(cfg)
    tol = dict(out_tol=5e-4, loss_tol=5e-5, grad_tol=1.5e-2, whole_tol=5e-3) if precise else \
        dict(out_tol=3e-2, loss_tol=1e-3, grad_tol=0.35, whole_tol=0.1, stat_tol=4e-2)
    check_with_injected_masks(backend, e, m, golden_sample(name, 0), **tol)


@pytest.mark.gpu
@pytest.mark.parametrize("precise", [True, False], ids=["bf16x3", "bf16"])
def test_base_recipe_dropouts_against_oracle(backend, precise):
    """BASELINE.json configs[1] -- the BENCHMARKED configuration (12 / 6 layers, d 512, 4 heads of 128: fused attention
    with dropout in fast mode, aux ASR / ST heads, CTC) with the recipe's dropouts (0.1 / 0.1 / 0.01, pre / post-net 0.5)
    on the base golden batch, against the oracle fed this step's masks: every loss term 5e-5 (bf16x3) / 1e-3 (bf16)."""
    if backend.kind != "hip":
        pytest.skip("base config runs on the GPU")
    cfg = dict(CONFIGS["base_recipe"])
    a, e = make_engine(backend, cfg, precise=precise)
    _, m = make_oracle(cfg)
    tol = dict(out_tol=5e-4, loss_tol=5e-5, grad_tol=1e-2, whole_tol=3e-3) if precise else \
        dict(out_tol=3e-2, loss_tol=1e-3, grad_tol=0.15, whole_tol=3e-2, stat_tol=4e-2)
    o, log = check_with_injected_masks(backend, e, m, golden_sample("base", 0), **tol)
    print(f"[base recipe dropouts {'bf16x3' if precise else 'bf16'}] loss {float(o['stats'][16]):.6f} vs oracle {float(log['loss']):.6f}")
