#!/usr/bin/env python3
"""Golden vectors of TEXT-INPUT generation and of the t2s encoder's SPEAKER PROJECTION from the reference (build container
only):
    python oracle/gen_golden_t2s_gen.py      # writes tests/golden/t2s_gen.npz, tests/golden/s2st_tiny_t2s_speaker.npz
TEST INFRASTRUCTURE.
(1) fairseq/speech_generator_for_s2st.py's AutoRegressiveSpeechGenerator(input_text=True) (:60-64: the encoder reads
    sample["src_text"] / ["src_text_len"]) on the reference's own t2s_transformer (tiny geometry, name-keyed synthetic
    weights, Prenet dropout 0): stop lengths, features, stop probabilities, alignments.
(2) The same model built with speakers -- the table from the reference mtl task's own ``get_speaker_embeddings(args)``
    (tasks/s2s_translation_mtl.py:133-150; it is the one task whose signature t2s_transformer.py:317 can call), i.e.
    Embedding(len(args.speaker_to_id) = length of the flag's STRING, speaker_embed_dim), and spk_emb_proj over
    cat[x, row] (t2s_transformer.py:43-46, 107-111): criterion forward / backward (losses, outputs, every gradient's
    norm + samples, the table's and the projection's gradients in full) and text-input AR generation with speakers.
The oracle must reproduce all of it before the files are written."""
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
from examples.s2s_trans.models.t2s_transformer import T2STransformerModel, base_architecture as t2s_arch  # noqa: E402
from examples.s2s_trans.criterions.t2s_loss import Tacotron2Criterion as T2SCriterion  # noqa: E402
from examples.s2s_trans.tasks.s2s_translation_mtl import S2ST_TranslationMTLTask as RefMTLTask  # noqa: E402
from fairseq.speech_generator_for_s2st import AutoRegressiveSpeechGenerator  # noqa: E402
import gen_golden as GG  # noqa: E402
from configs import CONFIGS, golden_sample  # noqa: E402
from synth_weights import load_synth  # noqa: E402
import s2st_oracle as O  # noqa: E402
import infer_oracle as IO  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
BASE_CFG = dict(CONFIGS["tiny_t2s"], prenet_dropout=0.0)
SPEAKER_TO_ID = '{"spk0": 0, "spk1": 1, "spk2": 2, "spk3": 3}'
SPK_CFG = dict(BASE_CFG, speaker_to_id=SPEAKER_TO_ID, speaker_embed_dim=24)
SPEAKER_IDS = [2, 0, 3, 1, 1, 2, 0, 3]
MAX_ITER, THR = 11, 0.235


def build(cfg):
    a = O.make_args(**cfg)
    ns = argparse.Namespace(**vars(a))
    t2s_arch(ns)
    src_d, tgt_d = GG.make_dict(a.src_vocab_size), GG.make_dict(a.tgt_vocab_size)

    class FakeTask:
        source_dictionary = src_d
        target_dictionary = tgt_d
        src_dict = src_d
        tgt_dict = tgt_d
        args = ns
        get_speaker_embeddings = RefMTLTask.get_speaker_embeddings  # the reference's own table builder (classmethod)

    ns.speaker_to_id = cfg.get("speaker_to_id")
    ns.speaker_emb_path = None
    model = T2STransformerModel.build_model(ns, FakeTask)
    load_synth(model, seed=0)
    return a, model, FakeTask


def pick_threshold(model, s):
    """A stop threshold that mixes early stops and max_iter and stays as far as possible from every stop probability the
    run compares with it (so that the integer outputs -- the stop indices -- do not hinge on rounding): probabilities of a
    never-stopping run (an utterance's values up to its own stop do not depend on the others'), thresholds on a grid."""
    class DC:
        tgt_global_cmvn_stats_npz = None
    model.eval()
    gen = AutoRegressiveSpeechGenerator(model, None, DC, max_iter=MAX_ITER, eos_prob_threshold=2.0, input_text=True)
    with torch.no_grad():
        fin = gen.generate(model, s)
    p = np.stack([f["eos_prob"].numpy()[::4] for f in fin]).astype(np.float64)
    best, best_score = (-1.0, None), (0, 0.0)
    for thr in np.arange(0.05, 0.95, 0.0005):
        lens, margin = [], 1.0
        for row in p:
            hit = np.nonzero(row > thr)[0]
            stop = int(hit[0]) if len(hit) else MAX_ITER - 1
            lens.append(stop + 1 if len(hit) else MAX_ITER)
            margin = min(margin, float(np.abs(row[: stop + 1] - thr).min()))
        # (three distinct lengths where the run offers them, else two: early stops and max_iter mixed either way)
        score = (min(len(set(lens)), 3), margin if margin >= 2e-3 else 0.0)
        if len(set(lens)) >= 2 and score > best_score:
            best_score, best = score, (margin, float(thr))
    assert best[1] is not None and best[0] >= 2e-3, best
    return best[1], best[0]


def gen_record(model, s, rec):
    class DC:
        tgt_global_cmvn_stats_npz = None
    thr, margin = pick_threshold(model, s)
    rec["margin"] = margin
    globals()["THR"] = thr
    model.eval()
    gen = AutoRegressiveSpeechGenerator(model, None, DC, max_iter=MAX_ITER, eos_prob_threshold=THR, input_text=True)
    with torch.no_grad():
        fin = gen.generate(model, s)
    for b, f in enumerate(fin):
        rec[f"feature.{b}"] = f["feature"].numpy()
        rec[f"eos_prob.{b}"] = f["eos_prob"].numpy()
        rec[f"alignment.{b}"] = f["alignment"].numpy()
        rec[f"attn.{b}"] = f["attn"].numpy()
    rec.update(n=len(fin), max_iter=MAX_ITER, thr=THR)
    return fin


def check_oracle_gen(cfg, s, fin, speaker=None):
    m = O.S2STModel(O.make_args(**cfg))
    load_synth(m, 0)
    mine = IO.ar_generate(m, s["src_text"], s["src_text_len"], MAX_ITER, THR, 4, speaker=speaker)
    for b, f in enumerate(fin):
        assert mine[b]["feature"].shape == f["feature"].shape, (b, mine[b]["feature"].shape, f["feature"].shape)
        assert float((mine[b]["feature"] - f["feature"]).abs().max()) < 2e-4
        assert torch.equal(mine[b]["alignment"], f["alignment"])
    return m


def sample(speaker=None):
    s = golden_sample("tiny", 0)
    s["net_input"]["collated_audios_orig"] = None
    s["net_input"]["padding_mask"] = None
    s["speaker"] = speaker
    return s


def main():
    # ---- (1) text-input generation, no speakers ------------------------------------------------------------------
    a, model, _ = build(BASE_CFG)
    s = sample()
    rec = {}
    fin = gen_record(model, s, rec)
    check_oracle_gen(BASE_CFG, s, fin)
    lens = [int(f["feature"].shape[0]) for f in fin]
    assert len(set(lens)) > 1, lens
    np.savez_compressed(os.path.join(OUT, "t2s_gen.npz"), **rec)
    print("t2s text-input generation golden: lens", lens)

    # ---- (2) speakers: training step + generation ------------------------------------------------------------------
    a, model, task = build(SPK_CFG)
    rows = model.encoder.embed_speaker.weight.shape[0]
    assert rows == len(SPEAKER_TO_ID) and model.encoder.embed_speaker.weight.shape[1] == 24
    assert tuple(model.encoder.spk_emb_proj.weight.shape) == (a.encoder_embed_dim, a.encoder_embed_dim + 24)
    spk = torch.tensor(SPEAKER_IDS, dtype=torch.long).view(-1, 1)
    s = sample(spk)
    crit = T2SCriterion(task, False, a.n_frames_per_step, False, 0.4, a.bce_pos_weight, 0.0)
    model.train()
    loss, ss, log = crit(model, s)
    loss.backward()
    rec = {"speaker_to_id": SPEAKER_TO_ID, "speaker_ids": np.array(SPEAKER_IDS), "rows": rows, "speaker_embed_dim": 24}
    for k, v in log.items():
        rec[f"log.{k}"] = np.asarray(float(v))
    named = dict(model.named_parameters())
    gn = {n: float(p.grad.norm()) for n, p in named.items() if p.grad is not None}
    rec["grad_norm_names"] = np.array(sorted(gn))
    rec["grad_norms"] = np.array([gn[k] for k in sorted(gn)], dtype=np.float64)
    for n in sorted(gn):
        rec[f"gsub.{n}"] = GG.gsub(GG.to_np(named[n].grad))
    for n in ("encoder.embed_speaker.weight", "encoder.spk_emb_proj.weight", "encoder.spk_emb_proj.bias"):
        rec["grad." + n] = named[n].grad.numpy().copy()
    sd = model.state_dict()
    rec["sd_names"] = np.array(list(sd.keys()))
    rec["sd_shapes"] = np.array([",".join(str(int(x)) for x in v.shape) for v in sd.values()])
    _, model2, _ = build(SPK_CFG)
    model2.train()
    with torch.no_grad():
        post, eos, extra = model2(src_tokens=s["src_text"], src_lengths=s["src_text_len"],
                                  prev_output_tokens=s["net_input"]["prev_output_tokens"], incremental_state=None,
                                  target_lengths=s["target_lengths"], speaker=spk)
        enc = model2.encoder(s["src_text"], s["src_text_len"], speaker=spk)
    for k, t in dict(post_feat_out=post, eos_out=eos, feature_out=extra["feature_out"], encoder_out=enc["encoder_out"][0]).items():
        rec[f"out.{k}"] = GG.to_np(t).astype(np.float32)
    # the oracle reproduces the training step
    m = O.S2STModel(O.make_args(**SPK_CFG))
    load_synth(m, 0)
    m.train()
    l2, _, lg2, _ = O.criterion_forward(m, s)
    l2.backward()
    assert abs(float(l2) - float(loss)) < 2e-5 * abs(float(loss)), (float(l2), float(loss))
    mine = dict(m.named_parameters())
    assert set(mine) == set(named), set(mine) ^ set(named)
    for n, p in named.items():
        if p.grad is None:
            continue
        d = float((mine[n].grad - p.grad).norm())
        assert d <= 2e-3 * float(p.grad.norm()) + 5e-6, (n, d, float(p.grad.norm()))  # (conv biases in front of BatchNorm: mathematically zero gradients, ~5e-7 of rounding noise on both sides)
    _, model3, _ = build(SPK_CFG)
    fin = gen_record(model3, s, rec)
    check_oracle_gen(SPK_CFG, s, fin, speaker=spk)
    np.savez_compressed(os.path.join(OUT, "s2st_tiny_t2s_speaker.npz"), **rec)
    print("t2s speaker golden: loss %.5f rows %d lens %s" % (float(loss), rows, [int(f["feature"].shape[0]) for f in fin]))


if __name__ == "__main__":
    main()
