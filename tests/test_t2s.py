"""``t2s_transformer`` / ``t2s_loss``: Transformer-TTS -- a text encoder front (token embedding,
3 x conv k5 + BatchNorm + ReLU, projection, alpha-scaled positions, post-LN layers) in front of the shared mel decoder.
Golden from the reference's own model and criterion (oracle/gen_golden_t2s.py): oracle pinned on CPU, HIP path on the
emulator (micro geometry, vs oracle) and on the GPU (tiny, vs the reference golden)."""
import importlib
import os

import numpy as np
import pytest
import torch

import s2st_oracle as O
from configs import CONFIGS, golden_sample
from synth_weights import load_synth
from test_engine import MICRO, check_gradient_direction, make_engine, make_oracle, rel

PKG = "speech-to-speech-translation_amd"
KEYS = ("loss", "l1_loss", "mse_loss", "eos_loss")


def test_oracle_against_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "s2st_tiny_t2s.npz"))
    a, m = make_oracle(CONFIGS["tiny_t2s"])
    assert set(m.state_dict().keys()) == set(z["sd_names"].tolist())
    s = golden_sample("tiny", 0)
    loss, ss, log, outs = O.criterion_forward(m, s)
    loss.backward()
    for k in KEYS:
        np.testing.assert_allclose(float(log[k]), float(z[f"log.{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    for k in ("post_feat_out", "eos_out", "feature_out", "attn", "encoder_out"):
        assert rel(outs[k], torch.from_numpy(z[f"out.{k}"])) < 3e-5, k
    check_gradient_direction({n: p.grad for n, p in m.named_parameters() if p.grad is not None}, z, 2e-3, 5e-4, tag="tiny")
    sd = m.state_dict()
    for k in z.files:
        if k.startswith("buf."):
            np.testing.assert_allclose(sd[k[4:]].numpy(), z[k], rtol=1e-4, atol=1e-6, err_msg=k)


def _engine_vs_oracle(backend, cfg, sample, precise, otol, gtol, ltol):
    a, e = make_engine(backend, cfg, precise=precise)
    _, m = make_oracle(cfg)
    loss, ss, log, outs = O.criterion_forward(m, sample)
    loss.backward()
    o = e.forward(sample, training=True, want_attn=True, seed=1)
    e.zero_grad()
    e.backward(1.0)
    backend.sync()
    for k, ref in (("encoder_out", outs["encoder_out"].transpose(0, 1)), ("feature_out", outs["feature_out"]),
                   ("eos_out", outs["eos_out"]), ("post_feat_out", outs["post_feat_out"]), ("attn", outs["attn"])):
        assert rel(o[k], ref) < otol, (k, rel(o[k], ref))
    st = o["stats"].cpu()
    for k, i in (("loss", 16), ("l1_loss", 17), ("mse_loss", 18), ("eos_loss", 19)):
        assert abs(float(st[i]) - float(log[k])) < ltol * max(1.0, abs(float(log[k]))), k
    named = dict(m.named_parameters())
    gmax = max(float(p.grad.norm()) for p in named.values() if p.grad is not None)
    for name, pv, gv, isb in e.named_views():
        if isb:
            continue
        rg = named[name].grad
        rg = torch.zeros_like(named[name]) if rg is None else rg
        d = float((gv.cpu() - rg).norm())
        assert d < gtol * (float(rg.norm()) + 1e-3 * gmax), (name, d, float(rg.norm()))
    # BatchNorm running statistics of the encoder prenet after one training forward
    bufs = dict((n, pv) for n, pv, _, b in e.named_views() if b)
    for n, v in m.state_dict().items():
        if "running_" in n:
            assert rel(bufs[n], v) < 2e-4, n
    return e, m


def test_micro_text_front_against_oracle(backend):
    D = importlib.import_module(PKG + ".data")
    cfg = dict(MICRO, asr_ce_weight=0.0, st_ce_weight=0.0, ctc_weight=0.0, text_encoder=True, encoder_conv_layers=2,
               encoder_conv_kernel_size=5, encoder_dropout=0.0, encoder_normalize_before=False)
    c = D.SyntheticFisherCorpus(n_utts=4, seed=3, max_src=64, median_src=50, min_src=30)
    _engine_vs_oracle(backend, cfg, c.collate_batch(range(4)), True, 3e-4, 1e-2, 3e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("precise", [True, False], ids=["bf16x3", "bf16"])
def test_tiny_t2s_through_task_model_criterion_against_reference_golden(backend, golden_dir, precise):
    if backend.kind != "hip":
        pytest.skip("tiny-size goldens run on the GPU")
    z = np.load(os.path.join(golden_dir, "s2st_tiny_t2s.npz"))
    tasks = importlib.import_module(PKG + ".tasks")
    a = O.make_args(**CONFIGS["tiny_t2s"])
    a.precise_gemm, a.arch, a.criterion = precise, "t2s_transformer", "t2s_loss"
    task = tasks.S2ST_TranslationTask.setup_task(a, device=backend.device)
    model = task.build_model(a)
    assert set(model.state_dict().keys()) == set(z["sd_names"].tolist())
    load_synth(model, 0)
    crit = task.build_criterion(a)
    model.train()
    s = golden_sample("tiny", 0)
    loss, ss, log = crit(model, s)
    model.engine.zero_grad()
    loss.backward()
    backend.sync()
    ltol = 5e-5 if precise else 1e-3
    for k in KEYS:
        r = float(z[f"log.{k}"])
        assert abs(float(log[k]) - r) < ltol * max(1.0, abs(r)), (k, float(log[k]), r)
    o = crit.last_outputs
    for k in ("post_feat_out", "eos_out", "feature_out"):
        assert rel(o[k], torch.from_numpy(z[f"out.{k}"])) < (3e-4 if precise else 3e-2), k
    assert rel(o["encoder_out"].transpose(0, 1), torch.from_numpy(z["out.encoder_out"])) < (3e-4 if precise else 3e-2)
    grads = {n: gv for n, pv, gv, isb in model.engine.named_views() if not isb}
    check_gradient_direction(grads, z, 1.5e-2 if precise else 1.5e-1, 5e-3 if precise else 5e-2, tag="tiny")
    if precise:
        assert np.array_equal(O.stop_indices(o["eos_out"].cpu()).numpy(), z["int.stop_idx"])
    # the model's own forward (FairseqEncoderDecoderModel signature of the reference)
    model.eval()
    post, eos, extra = model(s["src_text"], s["src_text_len"], s["net_input"]["prev_output_tokens"],
                             target_lengths=s["target_lengths"])
    assert post.shape == s["tgt_speech"].shape and extra["attn"].shape[0] == post.shape[0]


# ---- the feature-level CTC head (t2s_transformer.py:168-170, 258; t2s_loss.py:134-144) --------------------------------
def test_oracle_with_ctc_head_against_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "s2st_tiny_t2s_ctc.npz"))
    a, m = make_oracle(dict(CONFIGS["tiny_t2s"], ctc_weight=0.3))
    assert set(m.state_dict().keys()) == set(z["sd_names"].tolist())
    assert tuple(m.decoder.ctc_proj.weight.shape) == (a.src_vocab_size, 320)
    loss, ss, log, outs = O.criterion_forward(m, golden_sample("tiny", 0))
    loss.backward()
    for k in KEYS + ("ctc_loss",):
        np.testing.assert_allclose(float(log[k]), float(z[f"log.{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    check_gradient_direction({n: p.grad for n, p in m.named_parameters() if p.grad is not None}, z, 2e-3, 5e-4, tag="tiny")


def test_micro_text_front_with_ctc_head_against_oracle(backend):
    D = importlib.import_module(PKG + ".data")
    cfg = dict(MICRO, asr_ce_weight=0.0, st_ce_weight=0.0, ctc_weight=0.3, text_encoder=True, encoder_conv_layers=2,
               encoder_conv_kernel_size=5, encoder_dropout=0.0, encoder_normalize_before=False)
    c = D.SyntheticFisherCorpus(n_utts=4, seed=3, max_src=64, median_src=50, min_src=30)
    s = c.collate_batch(range(4))
    e, m = _engine_vs_oracle(backend, cfg, s, True, 3e-4, 1e-2, 3e-5)
    assert e.cfg.has_ctc == 1 and e.cfg.tap_asr == -1
    _, _, log, _ = O.criterion_forward(m, s)
    o = e.forward(s, training=True, seed=1)
    backend.sync()
    assert float(log["ctc_loss"]) > 0
    assert abs(float(o["stats"][20]) - float(log["ctc_loss"])) < 3e-5 * max(1.0, float(log["ctc_loss"]))


@pytest.mark.gpu
@pytest.mark.parametrize("precise", [True, False], ids=["bf16x3", "bf16"])
def test_tiny_t2s_with_ctc_head_against_reference_golden(backend, golden_dir, precise):
    if backend.kind != "hip":
        pytest.skip("tiny-size goldens run on the GPU")
    z = np.load(os.path.join(golden_dir, "s2st_tiny_t2s_ctc.npz"))
    tasks = importlib.import_module(PKG + ".tasks")
    a = O.make_args(**dict(CONFIGS["tiny_t2s"], ctc_weight=0.3))
    a.precise_gemm, a.arch, a.criterion = precise, "t2s_transformer", "t2s_loss"
    task = tasks.S2ST_TranslationTask.setup_task(a, device=backend.device)
    model = task.build_model(a)
    assert set(model.state_dict().keys()) == set(z["sd_names"].tolist())
    load_synth(model, 0)
    crit = task.build_criterion(a)
    model.train()
    loss, ss, log = crit(model, golden_sample("tiny", 0))
    model.engine.zero_grad()
    loss.backward()
    backend.sync()
    ltol = 5e-5 if precise else 1e-3
    for k in KEYS + ("ctc_loss",):
        r = float(z[f"log.{k}"])
        assert abs(float(log[k]) - r) < ltol * max(1.0, abs(r)), (k, float(log[k]), r)
    grads = {n: gv for n, pv, gv, isb in model.engine.named_views() if not isb}
    check_gradient_direction(grads, z, 1.5e-2 if precise else 1.5e-1, 5e-3 if precise else 5e-2, tag="tiny")
