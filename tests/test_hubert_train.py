"""BASELINE.json configs[3]: --use-hubert (frozen hubert_base front end) + the base model + aux ASR/ST decoders as ONE
training step (forward, s2st_loss, backward, clip, Adam), against tests/golden/s2st_hubert_train.npz -- produced by
the reference's own encoder HuBERT branch / model / criterion / Adam (oracle/gen_golden_hubert_train.py).

* CPU (`-m "not gpu"`): the oracle chain (HuBERT oracle -> model oracle -> criterion / optimizer restatement)
  against the golden, so the checker is pinned for this composition too.
* GPU (`-m gpu`): the HIP path through the host mirror (task -> model -> trainer), bf16x3 and bf16 modes.
"""
import importlib
import os

import numpy as np
import pytest
import torch

import hubert_oracle as HO
import s2st_oracle as O
from configs import CONFIGS, hubert_train_sample
from synth_weights import load_synth
from test_engine import LOSS_KEYS, bf16_tensor_bounds, check_gradient_direction

PKG = "speech-to-speech-translation_amd"
GEO = HO.HUBERT_CONFIGS["base"]


def _oracle_sample(s):
    ni = s["net_input"]
    with torch.no_grad():
        feats, fpm = HO.extract_features(HO.synth_state(GEO), GEO, ni["collated_audios_orig"], ni["padding_mask"])
    out = dict(s)
    out["net_input"] = dict(ni, src_speech=feats, src_speech_lens=(~fpm).long().sum(-1))
    return out, (~fpm).long().sum(-1)


def test_oracle_chain_against_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "s2st_hubert_train.npz"))
    a = O.make_args(**CONFIGS["hubert_train"])
    a._hubert_input = True
    m = O.S2STModel(a)
    load_synth(m, 0)
    m.train()
    s, frames = _oracle_sample(hubert_train_sample(0))
    assert np.array_equal(frames.numpy(), z["int.hubert_frames"])
    loss, ss, log, outs = O.criterion_forward(m, s)
    loss.backward()
    for k, _ in LOSS_KEYS:
        np.testing.assert_allclose(float(log[k]), float(z[f"log.{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    assert int(log["asr_n_correct"]) == int(z["log.asr_n_correct"]) and int(log["st_n_correct"]) == int(z["log.st_n_correct"])
    grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    check_gradient_direction(grads, z, 2e-3, 5e-4, tag="oracle")
    assert np.array_equal(O.stop_indices(outs["eos_out"]).numpy(), z["int.stop_idx"])


def _build(backend, precise):
    tasks = importlib.import_module(PKG + ".tasks")
    tr = importlib.import_module(PKG + ".trainer")
    a = O.make_args(**CONFIGS["hubert_train"])
    a.precise_gemm = precise
    a.report_accuracy = True
    a.lr, a.warmup_updates, a.clip_norm = float(1e-3), 2, 0.02
    task = tasks.S2ST_TranslationTask.setup_task(a, device=backend.device)
    model = task.build_model(a)  # hubert_base geometry is the default of --use-hubert true
    load_synth(model, 0)
    model.hubert.load_state_dict(HO.synth_state(GEO))
    crit = task.build_criterion(a)
    return a, task, model, crit, tr.Trainer(a, task, model, crit)


@pytest.mark.gpu
@pytest.mark.parametrize("precise", [True, False], ids=["bf16x3", "bf16"])
def test_hubert_base_training_step_golden(backend, golden_dir, precise):
    if backend.kind != "hip":
        pytest.skip("hubert_base + base model run on the GPU")
    z = np.load(os.path.join(golden_dir, "s2st_hubert_train.npz"))
    ltol, otol, gt, gw = (5e-5, 5e-4, 5e-3, 2e-3) if precise else (1e-3, 3e-2, 5e-2, 2e-2)
    a, task, model, crit, trainer = _build(backend, precise)
    eng = model.engine
    # ---- one forward + backward on batch 0 against the reference's tensors / gradients ----------------
    s = model.front_end_sample(hubert_train_sample(0))
    assert np.array_equal(model.hubert.last_frame_lens.numpy(), z["int.hubert_frames"])
    feats = s.hubert_io[2].double().cpu()  # the prepared batch's feature buffer, just refilled by the front end
    ref = z["sum.hubert_features"]
    assert abs(float(feats.abs().sum()) - ref[1]) < otol * ref[1]
    o = eng.forward(s, training=True, seed=1)
    eng.zero_grad()
    eng.backward(1.0)
    backend.sync()
    st = o["stats"].cpu()
    total = float(z["log.loss"])
    print("[hubert train %s] loss terms (mine, reference): " % ("bf16x3" if precise else "bf16") +
          ", ".join(f"{k} {float(st[i]):.5f}/{float(z['log.' + k]):.5f}" for k, i in LOSS_KEYS))
    for k, i in LOSS_KEYS:
        r = float(z[f"log.{k}"])
        if precise:
            assert abs(float(st[i]) - r) < ltol * max(1.0, abs(r)), (k, float(st[i]), r)
        else:
            # bf16 operands through the 12-layer frozen front end AND the model: the loss is held to the north-star
            # 1e-3 (relative), every term to 1e-3 of the loss (the BCE stop term, one logit per step, moves most)
            assert abs(float(st[i]) - r) < ltol * (abs(r) if k == "loss" else total), (k, float(st[i]), r)
    for k in ("post_feat_out", "feature_out", "eos_out", "asr_logits", "st_logits"):
        t = o[k].detach().cpu().double()
        r = z[f"sum.{k}"]
        # (bf16 mode: the stop logits are one 256 -> 1 projection per step with values ~0.1 at these synthetic
        # weights; their |sum| has been measured 2.6 ... 3.1 % off across the rounds' GEMM forms, everything else < 1 %)
        tol_k = 5e-2 if (k == "eos_out" and not precise) else otol
        assert abs(float(t.abs().sum()) - r[1]) < tol_k * r[1], k
        assert abs(float((t ** 2).sum().sqrt()) - r[2]) < tol_k * r[2], k
    if precise:
        assert np.array_equal(O.stop_indices(o["eos_out"].cpu()).numpy(), z["int.stop_idx"])
        assert int(st[5]) == int(z["log.asr_n_correct"]) and int(st[9]) == int(z["log.st_n_correct"])
    grads = {n: gv for n, pv, gv, isb in eng.named_views() if not isb}
    # bf16 mode: every tensor within 1.5 x of what the REFERENCE's own mixed precision (torch.autocast bf16 over the reference
    # model, the frozen HuBERT included: oracle/gen_golden_autocast.py) does to it on this batch -- whole gradient 1.8e-2,
    # first prenet layer 1.1e-1, post-net convolutions 7e-2 there; this path measured 1.6e-2 / 9.9e-2 / 5e-2
    ac_tol, ac_whole, ac = bf16_tensor_bounds(golden_dir, "s2st_hubert_train_autocast.npz")
    w, whole = check_gradient_direction(grads, z, gt if precise else ac_tol, gw if precise else ac_whole,
                                        tag="hubert " + ("bf16x3" if precise else "bf16"), yardstick=None if precise else ac)
    print(f"[hubert train {'bf16x3' if precise else 'bf16'}] worst tensor {w[1]} {w[0]:.2e}, whole gradient {whole:.2e}")
    bufs = dict((n, pv) for n, pv, _, b in eng.named_views() if b)
    for k in z.files:
        if k.startswith("buf."):
            np.testing.assert_allclose(bufs[k[4:]].cpu().numpy(), z[k], rtol=2e-4 if precise else 3e-2,
                                       atol=2e-5 if precise else 3e-3, err_msg=k)
    # ---- two optimizer updates through the trainer against the reference's Adam / clip -------------------
    a, task, model, crit, trainer = _build(backend, precise)
    for u in range(2):
        r = trainer.train_step([hubert_train_sample(u % 2)])
        backend.sync()
        np.testing.assert_allclose(float(r["logs"][0]["loss"]), z["train.loss"][u], rtol=1e-4 if precise else 1e-3)
        np.testing.assert_allclose(float(r["gnorm"]), z["train.gnorm"][u], rtol=1e-2 if precise else 5e-2)
    trainer.check_overflow()
    pn = dict(zip(z["train.param_norm_names"].tolist(), z["train.param_norms"].tolist()))
    for n, p in model.named_parameters():
        np.testing.assert_allclose(float(p.detach().norm()), pn[n], rtol=1e-4 if precise else 1e-3, err_msg=n)
