#!/usr/bin/env python3
"""Golden vectors of the mtl task's GENERATOR from the reference (build container only):
    python oracle/gen_golden_infer_mtl.py      # writes tests/golden/infer_mtl.npz
TEST INFRASTRUCTURE.  fairseq/speech_generator_for_s2st_mtl.py's AutoRegressiveSpeechGenerator.generate(model, sample,
decode_source_text=True, decode_target_mel=True) on the reference's own s2st_transformer_mtl model (tiny geometry, name-keyed
synthetic weights, Prenet dropout 0): per utterance the greedy CTC hypothesis (string and token ids), the reference string,
the corpus WER, and the AR mel outputs (stop lengths, features, stop probabilities, alignments).  The batch is the seeded
tiny batch in the mtl dataset's format (source text without EOS, ``source_texts``).  The CTC projection's bias is tilted
towards a few labels so that the hypotheses are not empty (random weights make "blank" or one label win everywhere
otherwise); the tilt is stored."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
for _n, _t in dict(float=float, int=int, bool=bool, object=object, complex=complex, str=str).items():
    if not hasattr(np, _n):
        setattr(np, _n, _t)
torch._C.has_cudnn = False
import fairseq  # noqa: E402,F401
from examples.s2s_trans.models.s2st_transformer_mtl import S2STTransformerModel as MTLModel, base_architecture as mtl_arch  # noqa: E402
from fairseq.speech_generator_for_s2st_mtl import AutoRegressiveSpeechGenerator  # noqa: E402
import gen_golden as GG  # noqa: E402
from configs import CONFIGS  # noqa: E402
from synth_weights import load_synth  # noqa: E402
import s2st_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
CFG = dict(CONFIGS["tiny_mtl"], prenet_dropout=0.0)
MAX_ITER, THR = 12, 0.235


def mtl_sample(src_d, tgt_d):
    """The seeded tiny batch through the package's mtl collater (tests pin that collater against the reference's)."""
    D = GG.D
    c = D.SyntheticFisherCorpus(n_utts=16, seed=1, max_src=200, median_src=120, mtl=True, src_dict=src_d, tgt_dict=tgt_d)
    return c.collate_batch(list(range(8)))


def ctc_bias_tilt(V):
    rs = np.random.RandomState(3)
    return torch.from_numpy((rs.standard_normal(V) * 1.5).astype(np.float32))


def main():
    a = O.make_args(**CFG)
    ns = argparse.Namespace(**vars(a))
    mtl_arch(ns)
    src_d, tgt_d = GG.make_dict(a.src_vocab_size), GG.make_dict(a.tgt_vocab_size)

    class FakeTask:
        source_dictionary = src_d
        target_dictionary = tgt_d
        src_dict = src_d
        tgt_dict = tgt_d
        args = ns

        @staticmethod
        def get_speaker_embeddings(args):
            return None

    ns.speaker_to_id = None
    model = MTLModel.build_model(ns, FakeTask)
    load_synth(model, seed=0)
    tilt = ctc_bias_tilt(a.src_vocab_size)
    with torch.no_grad():
        model.decoder.ctc_proj.bias.add_(tilt)
        model.decoder.ctc_proj.weight.mul_(6.0)  # frame-dependent winners: repeats, blanks and label changes all occur
    model.eval()

    class DC:
        tgt_global_cmvn_stats_npz = None

    gen = AutoRegressiveSpeechGenerator(model, None, DC, max_iter=MAX_ITER, eos_prob_threshold=THR)
    s = mtl_sample(src_d, tgt_d)
    fin = gen.generate(model, s, decode_source_text=True, decode_target_mel=True)
    # the frame-level best path, straight from the reference model (what the generator collapses)
    with torch.no_grad():
        enc = model.forward_encoder(s["net_input"]["src_speech"], s["net_input"]["src_speech_lens"], speaker=None)
        lp = torch.log_softmax(model.decoder.ctc_proj(enc["out_middle_layers"][0]).transpose(0, 1), dim=-1)
    best = lp.argmax(-1)
    top2 = lp.topk(2, dim=-1).values
    rec = {"n": len(fin), "max_iter": MAX_ITER, "thr": THR, "ctc_bias_tilt": tilt.numpy(), "ctc_weight_gain": 6.0,
           "enc_lens": enc["src_lengths"].numpy(), "best_path": best.numpy(),
           "best_margin": (top2[..., 0] - top2[..., 1]).numpy().astype(np.float32)}
    from fairseq import scoring
    wer = scoring.build_scorer("wer", src_d)
    for b, f in enumerate(fin):
        rec[f"src_text.{b}"] = np.asarray(f["src_texts"])
        rec[f"hyp_text.{b}"] = np.asarray(f["hyps_src_texts"])
        wer.add_string(f["src_texts"], f["hyps_src_texts"])
        rec[f"feature.{b}"] = f["feature"].numpy()
        rec[f"eos_prob.{b}"] = f["eos_prob"].numpy()
        rec[f"alignment.{b}"] = f["alignment"].numpy()
        rec[f"attn.{b}"] = f["attn"].numpy()
    rec["wer"] = np.asarray(wer.score())
    rec["wer_counts"] = np.asarray([wer.distance, wer.ref_length])
    n_tok = [len(str(rec[f"hyp_text.{b}"]).split()) for b in range(len(fin))]
    assert min(n_tok) >= 1 and len(set(n_tok)) > 1, n_tok
    lens = [int(f["feature"].shape[0]) for f in fin]
    assert len(set(lens)) > 1, lens
    np.savez_compressed(os.path.join(OUT, "infer_mtl.npz"), **rec)
    print("mtl generator golden: hyp lengths", n_tok, "mel lens", lens, "WER", float(rec["wer"]),
          "min best-path margin", float(rec["best_margin"].min()))


if __name__ == "__main__":
    main()
