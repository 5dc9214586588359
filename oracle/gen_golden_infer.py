#!/usr/bin/env python3
"""Golden vectors of the inference path from the REFERENCE classes (build container only):
    python oracle/gen_golden_infer.py      # writes tests/golden/infer_{ar,gl}.npz
TEST INFRASTRUCTURE.  (1) AutoRegressiveSpeechGenerator.generate on the tiny s2st_transformer with
name-keyed synthetic weights, Prenet dropout 0 (it is always on in the reference, i.e. random), no vocoder;
(2) GriffinLim.forward on a seeded magnitude spectrogram with numpy's global RNG seeded."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.argv = [sys.argv[0]]
import gen_golden as GG  # noqa: E402  (sets up the reference import path + shims)
from fairseq.speech_generator_for_s2st import AutoRegressiveSpeechGenerator  # noqa: E402
from fairseq.models.text_to_speech.vocoder import GriffinLim  # noqa: E402

import s2st_oracle as O  # noqa: E402
import infer_oracle as IO  # noqa: E402
from configs import CONFIGS, golden_sample  # noqa: E402
from synth_weights import load_synth  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
AR_CFG = dict(CONFIGS["tiny"], prenet_dropout=0.0)
MAX_ITER, THR = 14, 0.235


def ar_golden():
    a, model, crit = GG.build_reference(AR_CFG)
    load_synth(model, 0)
    model.eval()

    class DC:
        tgt_global_cmvn_stats_npz = None
    gen = AutoRegressiveSpeechGenerator(model, None, DC, max_iter=MAX_ITER, eos_prob_threshold=THR)
    s = golden_sample("tiny", 0)
    s["net_input"]["collated_audios_orig"] = None
    s["net_input"]["padding_mask"] = None
    s["speaker"] = None
    fin = gen.generate(model, s)
    rec = {"n": len(fin), "max_iter": MAX_ITER, "thr": THR}
    for b, f in enumerate(fin):
        rec[f"feature.{b}"] = f["feature"].numpy()
        rec[f"eos_prob.{b}"] = f["eos_prob"].numpy()
        rec[f"alignment.{b}"] = f["alignment"].numpy()
        rec[f"attn.{b}"] = f["attn"].numpy()
    # the oracle must reproduce it
    m = O.S2STModel(O.make_args(**AR_CFG))
    load_synth(m, 0)
    ni = s["net_input"]
    mine = IO.ar_generate(m, ni["src_speech"], ni["src_speech_lens"], MAX_ITER, THR, 4)
    for b, f in enumerate(fin):
        assert mine[b]["feature"].shape == f["feature"].shape, (b, mine[b]["feature"].shape, f["feature"].shape)
        assert float((mine[b]["feature"] - f["feature"]).abs().max()) < 2e-4
        assert torch.equal(mine[b]["alignment"], f["alignment"])
    np.savez_compressed(os.path.join(OUT, "infer_ar.npz"), **rec)
    print("AR golden: lens", [int(f["feature"].shape[0]) for f in fin])


def gl_golden():
    n_fft, win, hop, F_, T = 256, 200, 64, 129, 23
    rs = np.random.RandomState(5)
    spec = torch.from_numpy(np.abs(rs.randn(F_, T)).astype(np.float32))
    rec = {"spec": spec.numpy(), "n_fft": n_fft, "win": win, "hop": hop}
    for n_iter in (0, 4):
        gl = GriffinLim(n_fft, win, hop, n_iter)
        np.random.seed(11)
        wave = gl(spec)
        ang = IO.initial_angles((F_, T), np.random.RandomState(11))
        mine = IO.griffin_lim(spec, ang, n_fft, win, hop, n_iter)
        assert float((mine - wave).abs().max()) < 1e-4 * float(wave.abs().max()), n_iter
        rec[f"wave.{n_iter}"] = wave.numpy()
    rec["angles"] = IO.initial_angles((F_, T), np.random.RandomState(11))
    np.savez_compressed(os.path.join(OUT, "infer_gl.npz"), **rec)
    print("GL golden ok, wave len", rec["wave.4"].shape)


def dtw_golden():
    from examples.s2s_trans.tasks.s2s_translation import batch_dynamic_time_warping
    g = torch.Generator().manual_seed(17)
    d = torch.rand(3, 23, 31, generator=g)
    d[1, :, 7] = d[1, :, 8]  # exact ties exercise the first-minimum rule
    d[2, 5, :] = 0.25
    shapes = torch.tensor([[23, 31], [17, 20], [9, 31]])
    cum, bp, pm = batch_dynamic_time_warping(d, shapes)
    c2, b2, p2 = IO.dtw(d, shapes)
    assert torch.equal(bp, b2) and torch.equal(pm, p2) and torch.equal(cum, c2)
    np.savez_compressed(os.path.join(OUT, "infer_dtw.npz"), dist=d.numpy(), shapes=shapes.numpy(), cum=cum.numpy(),
                        backptr=bp.numpy(), pathmap=pm.numpy())
    print("DTW golden ok, path lengths", pm.sum(dim=(1, 2)).tolist())


if __name__ == "__main__":
    dtw_golden()
    gl_golden()
    ar_golden()
