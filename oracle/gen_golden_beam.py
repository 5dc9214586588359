#!/usr/bin/env python3
"""Golden vectors for aux ASR / ST beam decoding from the REFERENCE (build container only):
    python oracle/gen_golden_beam.py        # writes tests/golden/aux_beam.npz
TEST INFRASTRUCTURE.  Does what fairseq_cli/generate_for_s2st.py:107-111, 178-219 does: swaps ``model.decoder`` for the
aux ASR / ST decoder of the tiny reference model (synthetic weights) and runs the reference's own ``SequenceGenerator``
(beam 5, the recipe's setting: run_baseline.sh:185) on a seeded batch.  Stores, per head and utterance, the token ids
and scores of every returned hypothesis."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as GG  # noqa: E402
from configs import CONFIGS, golden_sample  # noqa: E402
from synth_weights import load_synth  # noqa: E402

from fairseq.sequence_generator import SequenceGenerator  # noqa: E402


def main():
    out = {}
    cfg = CONFIGS["tiny"]
    sample = golden_sample("tiny", 0)
    for which, beam, max_len_b in (("st", 5, 30), ("asr", 5, 30), ("st", 1, 12)):
        a, model, crit = GG.build_reference(cfg)
        load_synth(model, seed=0)
        model.eval()
        dec = model.aux_st_decoder if which == "st" else model.aux_asr_decoder
        d = dec.dictionary
        model.decoder = dec  # generate_for_s2st.py:107-111
        gen = SequenceGenerator([model], d, beam_size=beam, max_len_a=0, max_len_b=max_len_b, min_len=1)
        with torch.no_grad():
            hypos = gen.generate([model], sample)
        tag = f"{which}_b{beam}_m{max_len_b}"
        out[f"{tag}.n"] = np.array([len(h) for h in hypos])
        for i, hs in enumerate(hypos):
            for j, h in enumerate(hs):
                out[f"{tag}.{i}.{j}.tokens"] = h["tokens"].numpy()
                out[f"{tag}.{i}.{j}.score"] = np.asarray(float(h["score"]))
                out[f"{tag}.{i}.{j}.pos"] = h["positional_scores"].numpy()
        print(tag, [[h["tokens"].tolist() for h in hs[:2]] for hs in hypos[:2]], [round(float(hs[0]["score"]), 4) for hs in hypos])
    path = os.path.join(GG.ROOT, "tests", "golden", "aux_beam.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
