"""Speaker conditioning (examples/s2s_trans/models/s2st_transformer.py:203-206, 441-444; tables
tasks/s2s_translation.py:153-172): the utterance's speaker row is added to every encoder position before the dropout and
replaces the decoder's first input frame.  Golden from the reference's own model / criterion / generator with the tables
its task builds (oracle/gen_golden_speaker.py: 44 rows -- the reference sizes the tables by the LENGTH OF THE JSON STRING);
oracle pinned on CPU, HIP path through task -> model -> criterion -> generator against the golden (emulator and GPU)."""
import importlib
import os

import numpy as np
import pytest
import torch

import infer_oracle as IO
import s2st_oracle as O
from configs import CONFIGS, golden_sample
from synth_weights import load_synth

PKG = "speech-to-speech-translation_amd"
KEYS = ("l1_loss", "mse_loss", "eos_loss", "ctc_loss", "aux_asr_loss", "aux_st_loss")


def _cfg(z):
    return dict(CONFIGS["tiny"], speaker_to_id=str(z["speaker_to_id"]), speaker_embed_dim=128, speaker_embed_dim_dec=320)


def _sample(z):
    s = golden_sample("tiny", 0)
    spk = torch.from_numpy(z["speaker_ids"]).long().view(-1, 1)
    s["speaker"] = spk
    s["net_input"]["speaker"] = spk
    s["net_input"]["collated_audios_orig"] = None
    s["net_input"]["padding_mask"] = None
    return s


def test_oracle_against_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "s2st_tiny_speaker.npz"))
    m = O.S2STModel(O.make_args(**_cfg(z)))
    assert m.encoder.embed_speaker.weight.shape == (int(z["rows"]), 128) and int(z["rows"]) == len(str(z["speaker_to_id"]))
    load_synth(m, 0)
    m.train()
    s = _sample(z)
    loss, ss, log, outs = O.criterion_forward(m, s)
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) < 2e-5 * float(z["loss"])
    for k in KEYS:
        np.testing.assert_allclose(float(log[k]), float(z[f"log.{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    for n in ("encoder.embed_speaker.weight", "decoder.embed_speaker.weight"):
        g = dict(m.named_parameters())[n].grad.numpy()
        ref = z["grad." + n]
        assert np.abs(g - ref).max() <= 2e-3 * np.abs(ref).max(), n
        used = sorted(set(z["speaker_ids"].tolist()))
        assert np.abs(ref[[i for i in range(ref.shape[0]) if i not in used]]).max() == 0.0  # only the batch's speakers


@pytest.mark.parametrize("precise", [True, pytest.param(False, marks=pytest.mark.gpu)], ids=["bf16x3", "bf16"])
def test_speaker_conditioning_through_task_model_criterion_generator(backend, golden_dir, precise):
    if not precise and backend.kind != "hip":
        pytest.skip("bf16 mode at tiny size runs on the GPU")
    z = np.load(os.path.join(golden_dir, "s2st_tiny_speaker.npz"))
    tasks = importlib.import_module(PKG + ".tasks")
    gen_mod = importlib.import_module(PKG + ".speech_generator")
    a = O.make_args(**_cfg(z))
    a.precise_gemm = precise
    task = tasks.S2ST_TranslationTask.setup_task(a, device=backend.device)
    assert task.speaker_to_id == {"spk0": 0, "spk1": 1, "spk2": 2, "spk3": 3}
    model = task.build_model(a)
    sd = model.state_dict()
    assert tuple(sd["encoder.embed_speaker.weight"].shape) == (int(z["rows"]), 128)
    assert tuple(sd["decoder.embed_speaker.weight"].shape) == (int(z["rows"]), 320)
    load_synth(model, 0)
    crit = task.build_criterion(a)
    model.train()
    s = _sample(z)
    loss, ss, log = crit(model, s)
    model.engine.zero_grad()
    loss.backward()
    backend.sync()
    ltol = 5e-5 if precise else 1e-3
    assert abs(float(loss) - float(z["loss"])) < ltol * float(z["loss"])
    for k in KEYS:
        r = float(z[f"log.{k}"])
        assert abs(float(log[k]) - r) < ltol * max(1.0, abs(r)), (k, float(log[k]), r)
    grads = {n: gv.detach().cpu().numpy() for n, pv, gv, isb in model.engine.named_views() if not isb}
    gtol = 1.5e-2 if precise else 1.5e-1
    for n in ("encoder.embed_speaker.weight", "decoder.embed_speaker.weight", "decoder.prenet.0.layers.0.0.weight",
              "encoder.subsample.conv_layers.1.bias"):
        ref = z["grad." + n]
        assert np.linalg.norm(grads[n] - ref) <= gtol * np.linalg.norm(ref), (n, np.linalg.norm(grads[n] - ref), np.linalg.norm(ref))
    names, norms = z["grad_names"].tolist(), z["grad_norms"]
    for n, r in zip(names, norms):
        if n in grads and r > 1e-6 and not n.endswith("k_proj.bias"):
            assert abs(np.linalg.norm(grads[n]) - r) <= (5e-2 if precise else 2.5e-1) * r + 1e-7, n
    # generator: the speaker row is the decoder's input at EVERY step (the reference's incremental path, reproduced)
    gen = gen_mod.AutoRegressiveSpeechGenerator(model, None, None, max_iter=int(z["max_iter"]), eos_prob_threshold=float(z["thr"]))
    fin = gen.generate(model, s)
    backend.sync()
    for b in range(int(z["n"])):
        ref = z[f"feature.{b}"]
        got = fin[b]["feature"].cpu().numpy()
        assert got.shape == ref.shape
        assert float(np.abs(got - ref).max()) < (5e-4 if precise else 3e-2) * max(1.0, float(np.abs(ref).max())), b
        if precise:
            assert np.array_equal(fin[b]["alignment"].cpu().numpy(), z[f"alignment.{b}"])


def test_flag_checks(backend):
    tasks = importlib.import_module(PKG + ".tasks")
    a = O.make_args(**dict(CONFIGS["tiny"], speaker_to_id='{"a": 0}'))  # default widths (64) fit neither use
    a.precise_gemm = True
    task = tasks.S2ST_TranslationTask.setup_task(a, device=backend.device)
    with pytest.raises(ValueError, match="speaker-embed-dim"):
        task.build_model(a)


def test_frozen_tables_stay_out_of_the_optimizer(backend, tmp_path):
    """tables from ``speaker_emb_path`` are ``Embedding.from_pretrained(freeze=True)`` in the reference
    (tasks/s2s_translation.py:161-171) -- requires_grad False, hence never handed to the optimizer.  Here they live in the
    engine's BUFFER arena: not a module parameter, no gradient, and Adam updates with weight decay leave them bit for
    bit -- while trainable tables of the same geometry move.  (One file serves both tables in the reference, so the
    encoder is as wide as a packed output frame here.)"""
    tasks = importlib.import_module(PKG + ".tasks")
    trainer_mod = importlib.import_module(PKG + ".trainer")
    from test_engine import MICRO
    D = importlib.import_module(PKG + ".data")
    spk = '{"a": 0, "b": 1}'
    c = D.SyntheticFisherCorpus(n_utts=4, seed=3, max_src=64, median_src=50, min_src=30)
    s = c.collate_batch(range(4))
    s["speaker"] = torch.tensor([3, 0, 3, 7]).view(-1, 1)
    s["net_input"]["speaker"] = s["speaker"]
    moved = {}
    for frozen in (True, False):
        cfg = dict(MICRO, encoder_embed_dim=320, speaker_to_id=spk, speaker_embed_dim=320, speaker_embed_dim_dec=320,
                   encoder_transformer_layers=1, decoder_transformer_layers=1, encoder_ffn_embed_dim=64,
                   decoder_ffn_embed_dim=64, middle_layers="0,0", asr_ce_weight=0.0, st_ce_weight=0.0, ctc_weight=0.0)
        a = O.make_args(**cfg)
        a.precise_gemm = True
        a.weight_decay, a.lr, a.warmup_updates, a.clip_norm = 0.1, [1e-2], 1, 1.0
        task = tasks.S2ST_TranslationTask.setup_task(a, device=backend.device)
        if frozen:
            path = str(tmp_path / "spk.npy")
            np.save(path, np.random.RandomState(0).standard_normal((len(spk), 320)).astype(np.float32))
            task.get_speaker_embeddings_path = lambda: path
        model = task.build_model(a)
        load_synth(model, 0)
        names_p = {n for n, _ in model.named_parameters()}
        names_b = {n for n, _ in model.named_buffers()}
        tabs = ["encoder.embed_speaker.weight", "decoder.embed_speaker.weight"]
        assert all((t in names_b) == frozen and (t in names_p) == (not frozen) for t in tabs)
        assert all(t in model.state_dict() for t in tabs)  # the checkpoint key stays either way
        before = {t: model._views[t].clone() for t in tabs}
        tr = trainer_mod.Trainer(a, task, model, task.build_criterion(a))
        tr.train_step([s])  # (the first update runs at lr 0: linear warm-up from 0)
        tr.train_step([s])
        backend.sync()
        moved[frozen] = {t: float((model._views[t] - before[t]).abs().max()) for t in tabs}
    assert all(v > 0 for v in moved[False].values()), moved
    assert all(v == 0.0 for v in moved[True].values()), moved
