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


# ---- text-input AR generation (speech_generator_for_s2st.py:60-64) and the encoder's speaker projection
#      (t2s_transformer.py:43-46, 107-111); goldens: oracle/gen_golden_t2s_gen.py ---------------------------------------
GEN_CFG = dict(CONFIGS["tiny_t2s"], prenet_dropout=0.0)
SPK_JSON = '{"spk0": 0, "spk1": 1, "spk2": 2, "spk3": 3}'
SPK_CFG = dict(GEN_CFG, speaker_to_id=SPK_JSON, speaker_embed_dim=24)


def _t2s_sample(z=None):
    s = golden_sample("tiny", 0)
    s["net_input"]["collated_audios_orig"] = None
    s["net_input"]["padding_mask"] = None
    s["speaker"] = None if z is None else torch.from_numpy(z["speaker_ids"]).long().view(-1, 1)
    return s


def _check_generated(fin, z, backend):
    lens = []
    for b in range(int(z["n"])):
        ref = z[f"feature.{b}"]
        assert tuple(fin[b]["feature"].shape) == ref.shape, (b, fin[b]["feature"].shape, ref.shape)  # the stop index
        lens.append(ref.shape[0])
        assert float(np.abs(fin[b]["feature"].cpu().numpy() - ref).max()) < 1e-3 * max(1.0, np.abs(ref).max())
        assert float(np.abs(fin[b]["eos_prob"].cpu().numpy() - z[f"eos_prob.{b}"]).max()) < 2e-4
        assert float(np.abs(fin[b]["attn"].cpu().numpy() - z[f"attn.{b}"]).max()) < 2e-4
        assert np.array_equal(fin[b]["alignment"].cpu().numpy(), z[f"alignment.{b}"])  # integer: bit-exact
    assert len(set(lens)) > 1 and float(z["margin"]) >= 2e-3


def _t2s_task_model(backend, cfg, precise=True, input_text="true"):
    tasks = importlib.import_module(PKG + ".tasks")
    a = O.make_args(**cfg)
    a.precise_gemm, a.arch, a.criterion, a.input_text = precise, "t2s_transformer", "t2s_loss", input_text
    task = tasks.S2ST_TranslationTask.setup_task(a, device=backend.device)
    model = task.build_model(a)
    load_synth(model, 0)
    return a, task, model


def test_text_input_generator_against_reference_golden(backend, golden_dir):
    z = np.load(os.path.join(golden_dir, "t2s_gen.npz"))
    a, task, model = _t2s_task_model(backend, GEN_CFG)
    gen_mod = importlib.import_module(PKG + ".speech_generator")
    gen = gen_mod.AutoRegressiveSpeechGenerator(model, None, None, max_iter=int(z["max_iter"]),
                                                eos_prob_threshold=float(z["thr"]), input_text=True)
    fin = gen.generate(model, _t2s_sample())
    backend.sync()
    _check_generated(fin, z, backend)
    # the task wires --input-text through (tasks/s2s_translation.py:186-204)
    assert task.build_generator_tts([model], a, vocoder=False).input_text is True
    with pytest.raises(ValueError):  # a text-input model decodes from text only, and only it does
        gen_mod.AutoRegressiveSpeechGenerator(model, None, None, input_text=False).generate(model, _t2s_sample())


def test_oracle_with_speaker_projection_against_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "s2st_tiny_t2s_speaker.npz"))
    a, m = make_oracle(SPK_CFG)
    assert set(m.state_dict().keys()) == set(z["sd_names"].tolist())
    assert tuple(m.encoder.embed_speaker.weight.shape) == (int(z["rows"]), 24) and int(z["rows"]) == len(SPK_JSON)
    loss, ss, log, outs = O.criterion_forward(m, _t2s_sample(z))
    loss.backward()
    for k in KEYS:
        np.testing.assert_allclose(float(log[k]), float(z[f"log.{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    named = dict(m.named_parameters())
    for n in ("encoder.embed_speaker.weight", "encoder.spk_emb_proj.weight", "encoder.spk_emb_proj.bias"):
        assert rel(named[n].grad, torch.from_numpy(z["grad." + n])) < 2e-4, n
    check_gradient_direction({n: p.grad for n, p in named.items() if p.grad is not None}, z, 2e-3, 5e-4, tag="tiny")


def test_micro_text_front_with_speaker_projection_against_oracle(backend):
    D = importlib.import_module(PKG + ".data")
    cfg = dict(MICRO, asr_ce_weight=0.0, st_ce_weight=0.0, ctc_weight=0.0, text_encoder=True, encoder_conv_layers=2,
               encoder_conv_kernel_size=5, encoder_dropout=0.0, encoder_normalize_before=False,
               speaker_to_id='{"a": 0, "b": 1}', speaker_embed_dim=12)
    c = D.SyntheticFisherCorpus(n_utts=4, seed=3, max_src=64, median_src=50, min_src=30)
    s = c.collate_batch(range(4))
    s["speaker"] = torch.tensor([3, 0, 3, 7]).view(-1, 1)  # (rows = len of the JSON string = 16; two utterances share a row)
    e, m = _engine_vs_oracle(backend, cfg, s, True, 3e-4, 1e-2, 3e-5)
    assert e.cfg.n_speakers == 16 and e.cfg.spk_dim == 12
    g = dict((n, gv) for n, _, gv, b in e.named_views() if not b)["encoder.embed_speaker.weight"].cpu()
    assert float(g[[3, 0, 7]].abs().sum()) > 0 and float(g[[1, 2, 4, 5, 6] + list(range(8, 16))].abs().sum()) == 0.0
    with pytest.raises(IndexError):  # an id outside the table is refused on the host (nn.Embedding would raise too)
        e.forward(dict(s, speaker=torch.tensor([0, 1, 16, 2]).view(-1, 1)), training=True, seed=1)
    with pytest.raises(ValueError):  # the reference's encoder embeds `speaker` unconditionally once the table exists
        e.forward(dict(s, speaker=None), training=True, seed=1)


@pytest.mark.gpu
@pytest.mark.parametrize("precise", [True, False], ids=["bf16x3", "bf16"])
def test_tiny_t2s_with_speakers_against_reference_golden(backend, golden_dir, precise):
    if backend.kind != "hip":
        pytest.skip("tiny-size goldens run on the GPU")
    z = np.load(os.path.join(golden_dir, "s2st_tiny_t2s_speaker.npz"))
    a, task, model = _t2s_task_model(backend, SPK_CFG, precise)
    assert set(model.state_dict().keys()) == set(z["sd_names"].tolist())
    crit = task.build_criterion(a)
    model.train()
    s = _t2s_sample(z)
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
    for n in ("encoder.embed_speaker.weight", "encoder.spk_emb_proj.weight", "encoder.spk_emb_proj.bias"):
        assert rel(grads[n], torch.from_numpy(z["grad." + n])) < (2e-3 if precise else 8e-2), (n, rel(grads[n], torch.from_numpy(z["grad." + n])))
    if precise:  # text-input generation WITH speakers (the table enters through the encoder only)
        # (a fresh model, as in the golden: the training forward above moved the BatchNorm running statistics)
        a, task, model = _t2s_task_model(backend, SPK_CFG, precise)
        gen = task.build_generator_tts([model], a, vocoder=False)
        gen.max_iter, gen.eos_prob_threshold = int(z["max_iter"]), float(z["thr"])
        fin = gen.generate(model, s)
        backend.sync()
        _check_generated(fin, z, backend)
