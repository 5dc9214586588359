"""``s2st_transformer_mtl`` / ``s2st_loss_mtl`` / ``s2s_translation_mtl``: the variant's
addition is a second CTC head -- target text, on the raw output of a decoder layer.  Golden from the reference's own
model and criterion (oracle/gen_golden_mtl.py): oracle pinned on CPU, HIP path on the emulator (micro, vs oracle) and
on the GPU (tiny, vs the reference golden)."""
import importlib
import os

import numpy as np
import pytest
import torch

import s2st_oracle as O
from configs import CONFIGS, golden_sample
from synth_weights import load_synth
from test_engine import MICRO, check_against_oracle, check_gradient_direction, make_engine, make_oracle, rel

PKG = "speech-to-speech-translation_amd"
KEYS = ("loss", "l1_loss", "mse_loss", "eos_loss", "ctc_loss", "ctc_loss_tgt")


def test_oracle_against_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "s2st_tiny_mtl.npz"))
    a, m = make_oracle(CONFIGS["tiny_mtl"])
    assert set(m.state_dict().keys()) == set(z["sd_names"].tolist())
    s = golden_sample("tiny", 0)
    loss, ss, log, outs = O.criterion_forward(m, s)
    loss.backward()
    for k in KEYS:
        np.testing.assert_allclose(float(log[k]), float(z[f"log.{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    for k in ("post_feat_out", "eos_out", "feature_out"):
        assert rel(outs[k], torch.from_numpy(z[f"out.{k}"])) < 2e-5, k
    check_gradient_direction({n: p.grad for n, p in m.named_parameters() if p.grad is not None}, z, 2e-3, 5e-4, tag="tiny")


def test_micro_engine_with_target_ctc_head_against_oracle(backend):
    D = importlib.import_module(PKG + ".data")
    cfg = dict(MICRO, asr_ce_weight=0.0, st_ce_weight=0.0, middle_layers="0", middle_layers_decoder="1", ctc_weight=0.3,
               ctc_weight_tgt=0.2)
    a, e = make_engine(backend, cfg, precise=True)
    assert e.cfg.has_ctc_tgt == 1 and e.cfg.tap_dec == 1
    _, m = make_oracle(cfg)
    c = D.SyntheticFisherCorpus(n_utts=4, seed=3, max_src=64, median_src=50, min_src=30)
    s = c.collate_batch(range(4))
    o, outs, log = check_against_oracle(backend, e, m, s, out_tol=3e-4, grad_tol=1e-2, loss_tol=3e-5)
    assert abs(float(o["stats"][23]) - float(log["ctc_loss_tgt"])) < 3e-5 * max(1.0, float(log["ctc_loss_tgt"]))
    assert float(log["ctc_loss_tgt"]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("precise", [True, False], ids=["bf16x3", "bf16"])
def test_tiny_mtl_through_task_model_criterion_against_reference_golden(backend, golden_dir, precise):
    if backend.kind != "hip":
        pytest.skip("tiny-size goldens run on the GPU")
    z = np.load(os.path.join(golden_dir, "s2st_tiny_mtl.npz"))
    tasks = importlib.import_module(PKG + ".tasks")
    a = O.make_args(**CONFIGS["tiny_mtl"])
    a.precise_gemm = precise
    task = tasks.S2ST_TranslationMTLTask.setup_task(a, device=backend.device)
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
        assert abs(float(log[k]) - r) < ltol * max(1.0, abs(r) if k != "loss" or precise else abs(r)), (k, float(log[k]), r)
    assert "aux_asr_loss" not in dict(log.items())
    o = crit.last_outputs
    for k in ("post_feat_out", "eos_out", "feature_out"):
        assert rel(o[k], torch.from_numpy(z[f"out.{k}"])) < (3e-4 if precise else 3e-2), k
    grads = {n: gv for n, pv, gv, isb in model.engine.named_views() if not isb}
    check_gradient_direction(grads, z, 1.5e-2 if precise else 1.5e-1, 5e-3 if precise else 5e-2, tag="tiny")
    if precise:
        assert np.array_equal(O.stop_indices(o["eos_out"].cpu()).numpy(), z["int.stop_idx"])
        # the model-level entry the reference's criterion uses for this head
        dec_tap_missing = pytest.raises(ValueError) if not model.engine.cfg.has_ctc_tgt else None
        assert dec_tap_missing is None
