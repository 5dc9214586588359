#!/usr/bin/env python3
"""Golden vectors of config 5 AT ITS STATED SIZE from the REFERENCE generator (build container only):
    python oracle/gen_golden_infer_base.py      # writes tests/golden/infer_ar_base.npz
TEST INFRASTRUCTURE.  AutoRegressiveSpeechGenerator.generate (fairseq/speech_generator_for_s2st.py:46-134) on the BASE
s2st_transformer (12 / 6 layers, d 512, n_frames_per_step 4) with name-keyed synthetic weights, Prenet dropout 0 (always on
in the reference, i.e. random), no vocoder, 8 Fisher-shaped utterances, max_iter = the longest teacher length.  The stop
threshold is chosen from the reference's own stop probabilities (a first pass that never stops) so that the batch mixes
early stops with utterances that run to max_iter.  The oracle must reproduce the result before it is written."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.argv = [sys.argv[0]]
import gen_golden as GG  # noqa: E402  (sets up the reference import path + shims)
from fairseq.speech_generator_for_s2st import AutoRegressiveSpeechGenerator  # noqa: E402

import s2st_oracle as O  # noqa: E402
import infer_oracle as IO  # noqa: E402
from configs import CONFIGS, golden_sample  # noqa: E402
from synth_weights import load_synth  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
AR_CFG = dict(CONFIGS["base"], prenet_dropout=0.0)


def main():
    torch.set_num_threads(8)
    a, model, crit = GG.build_reference(AR_CFG)
    load_synth(model, 0)
    model.eval()

    class DC:
        tgt_global_cmvn_stats_npz = None
    s = golden_sample("base", 0)
    s["net_input"]["collated_audios_orig"] = None
    s["net_input"]["padding_mask"] = None
    s["speaker"] = None
    max_iter = int(s["target_lengths"].max())
    # pass 1: never stop -> the reference's own stop probabilities
    gen = AutoRegressiveSpeechGenerator(model, None, DC, max_iter=max_iter, eos_prob_threshold=2.0)
    with torch.no_grad():
        fin = gen.generate(model, s)
    probs = torch.stack([f["eos_prob"][::4][:max_iter] for f in fin])  # [B, steps]
    # threshold: inside the band where between a quarter and three quarters of the utterances cross it before the last
    # step, at the centre of the WIDEST gap between two consecutive stop probabilities of the whole run -- the decision
    # must not hinge on the last digits (the bf16 path's stop probabilities move by ~2.5e-3 at this size)
    first_max = probs[:, :-1].max(dim=1).values.sort().values
    lo, hi = float(first_max[1]), float(first_max[-2])  # (at least two utterances stop early, at least two do not)
    vals = probs.flatten().sort().values
    gaps = [(float(vals[i + 1] - vals[i]), float(vals[i] + vals[i + 1]) / 2) for i in range(len(vals) - 1)
            if lo <= float(vals[i]) and float(vals[i + 1]) <= hi]
    gap, thr = max(gaps)
    assert gap / 2 >= 4e-3, gap
    gen = AutoRegressiveSpeechGenerator(model, None, DC, max_iter=max_iter, eos_prob_threshold=thr)
    with torch.no_grad():
        fin = gen.generate(model, s)
    lens = [int(f["feature"].shape[0]) for f in fin]
    assert len(set(lens)) > 1, lens
    rec = {"n": len(fin), "max_iter": max_iter, "thr": thr, "margin": float((probs - thr).abs().min())}
    for b, f in enumerate(fin):
        rec[f"feature.{b}"] = f["feature"].numpy()
        rec[f"eos_prob.{b}"] = f["eos_prob"].numpy()
        rec[f"alignment.{b}"] = f["alignment"].numpy()
        rec[f"attn_sum.{b}"] = f["attn"].double().sum(dim=0).float().numpy()  # [T] column masses of the alignment map
    m = O.S2STModel(O.make_args(**AR_CFG))
    load_synth(m, 0)
    ni = s["net_input"]
    with torch.no_grad():
        mine = IO.ar_generate(m, ni["src_speech"], ni["src_speech_lens"], max_iter, thr, 4)
    for b, f in enumerate(fin):
        assert mine[b]["feature"].shape == f["feature"].shape, (b, mine[b]["feature"].shape, f["feature"].shape)
        err = float((mine[b]["feature"] - f["feature"]).abs().max())
        assert err < 5e-4 * max(1.0, float(f["feature"].abs().max())), (b, err)
        assert torch.equal(mine[b]["alignment"], f["alignment"]), b
    np.savez_compressed(os.path.join(OUT, "infer_ar_base.npz"), **rec)
    print("base AR golden: thr %.4f margin %.2e lens %s max_iter %d" % (thr, rec["margin"], lens, max_iter))


if __name__ == "__main__":
    main()
