#!/usr/bin/env python3
"""Golden vectors of SPEAKER CONDITIONING from the reference (build container only):
    python oracle/gen_golden_speaker.py      # writes tests/golden/s2st_tiny_speaker.npz
TEST INFRASTRUCTURE.  The reference's own model (examples/s2s_trans/models/s2st_transformer.py:203-206, 441-444) with the
tables its task builds (tasks/s2s_translation.py:153-172 -- ``Embedding(len(args.speaker_to_id), dim)``, i.e. as many rows as
the JSON STRING has characters), the tiny geometry, name-keyed synthetic weights, one seeded batch with speaker ids in
``sample["speaker"]``: criterion forward / backward (losses, outputs, every gradient's norm, the two tables' gradients in
full) and the autoregressive generator (stop lengths, features) -- in which the speaker row replaces the one input frame at
EVERY step (the generator hands the decoder a single frame, :441-444 keeps ``prev[:, 1:]`` = nothing).  The oracle must
reproduce all of it before the file is written."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.argv = [sys.argv[0]]
import gen_golden as GG  # noqa: E402  (sets up the reference import path + shims)
from examples.s2s_trans.models.s2st_transformer import S2STTransformerModel, base_architecture  # noqa: E402
from examples.s2s_trans.criterions.s2st_loss import Tacotron2Criterion  # noqa: E402
import fairseq.tasks as _ft  # noqa: E402
# (gen_golden imported this package, whose plugin registered the task name first: free the name so that the reference's
# own task module -- which holds the table builder pinned here -- can be imported)
_ft.TASK_REGISTRY.pop("s2s_translation", None)
_ft.TASK_CLASS_NAMES.discard("S2ST_TranslationTask")
from examples.s2s_trans.tasks.s2s_translation import S2ST_TranslationTask as RefTask  # noqa: E402
from fairseq.speech_generator_for_s2st import AutoRegressiveSpeechGenerator  # noqa: E402

import s2st_oracle as O  # noqa: E402
import infer_oracle as IO  # noqa: E402
from configs import CONFIGS, golden_sample  # noqa: E402
from synth_weights import load_synth  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SPEAKER_TO_ID = '{"spk0": 0, "spk1": 1, "spk2": 2, "spk3": 3}'
CFG = dict(CONFIGS["tiny"], speaker_to_id=SPEAKER_TO_ID, speaker_embed_dim=128, speaker_embed_dim_dec=320)
SPEAKER_IDS = [2, 0, 3, 1, 1, 2, 0, 3]
MAX_ITER, THR = 9, 2.0


def build_reference(cfg):
    a = O.make_args(**cfg)
    ns = argparse.Namespace(**vars(a))
    base_architecture(ns)
    src_d, tgt_d = GG.make_dict(a.src_vocab_size), GG.make_dict(a.tgt_vocab_size)

    class FakeTask:
        source_dictionary = src_d
        target_dictionary = tgt_d
        src_dict = src_d
        tgt_dict = tgt_d
        args = ns
        get_speaker_embeddings = RefTask.get_speaker_embeddings  # the reference's own table builder (classmethod)

    ns.speaker_emb_path = None
    model = S2STTransformerModel.build_model(ns, FakeTask)
    crit = Tacotron2Criterion(
        FakeTask, sentence_avg=False, n_frames_per_step=a.n_frames_per_step, use_guided_attention_loss=False,
        guided_attention_loss_sigma=0.4, bce_pos_weight=a.bce_pos_weight, ctc_weight=a.ctc_weight,
        asr_ce_weight=a.asr_ce_weight, st_ce_weight=a.st_ce_weight, l1_loss_weight=a.l1_loss_weight,
        mse_loss_weight=a.mse_loss_weight, eos_loss_weight=a.eos_loss_weight, attn_loss_weight=a.attn_loss_weight,
        label_smoothing=a.label_smoothing, report_accuracy=True)
    return a, model, crit


def main():
    torch.manual_seed(0)
    a, model, crit = build_reference(CFG)
    load_synth(model, 0)
    rows = model.encoder.embed_speaker.weight.shape[0]
    assert rows == len(SPEAKER_TO_ID) and model.decoder.embed_speaker.weight.shape == (rows, 320)
    s = golden_sample("tiny", 0)
    spk = torch.tensor(SPEAKER_IDS, dtype=torch.long).view(-1, 1)
    s["speaker"] = spk
    s["net_input"]["speaker"] = spk
    s["net_input"]["collated_audios_orig"] = None
    s["net_input"]["padding_mask"] = None
    model.train()
    loss, ss, log = crit(model, s)
    loss.backward()
    rec = {"speaker_to_id": SPEAKER_TO_ID, "speaker_ids": np.array(SPEAKER_IDS), "rows": rows, "loss": float(loss),
           "max_iter": MAX_ITER, "thr": THR}
    for k in ("l1_loss", "mse_loss", "eos_loss", "ctc_loss", "aux_asr_loss", "aux_st_loss"):
        rec["log." + k] = float(log[k])
    with torch.no_grad():
        model.eval()
        ni = s["net_input"]
        (post, eos, extra), _, _ = model(ni["src_speech"], ni["src_speech_lens"], None, None, ni["prev_output_tokens"],
                                         prev_src_text_tokens=ni["prev_src_text_tokens"],
                                         prev_tgt_text_tokens=ni["prev_tgt_text_tokens"], incremental_state=None,
                                         target_lengths=s["target_lengths"], speaker=spk)
        model.train()
    names = [n for n, p in model.named_parameters()]
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array([float(p.grad.norm()) if p.grad is not None else 0.0 for _, p in model.named_parameters()])
    for n in ("encoder.embed_speaker.weight", "decoder.embed_speaker.weight", "decoder.prenet.0.layers.0.0.weight",
              "encoder.subsample.conv_layers.1.bias"):
        rec["grad." + n] = dict(model.named_parameters())[n].grad.numpy().copy()
    # the oracle reproduces the training step
    m = O.S2STModel(O.make_args(**CFG))
    load_synth(m, 0)
    m.train()
    l2, _, lg2, _ = O.criterion_forward(m, s)
    l2.backward()
    assert abs(float(l2) - float(loss)) < 2e-5 * abs(float(loss)), (float(l2), float(loss))
    mine = dict(m.named_parameters())
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        d = float((mine[n].grad - p.grad).norm())
        assert d <= 2e-3 * float(p.grad.norm()) + 1e-7, (n, d, float(p.grad.norm()))
    # generator
    model.eval()

    class DC:
        tgt_global_cmvn_stats_npz = None
    gen = AutoRegressiveSpeechGenerator(model, None, DC, max_iter=MAX_ITER, eos_prob_threshold=THR)
    with torch.no_grad():
        fin = gen.generate(model, s)
        got = IO.ar_generate(m, ni["src_speech"], ni["src_speech_lens"], MAX_ITER, THR, 4, speaker=spk)
    for b, f in enumerate(fin):
        rec[f"feature.{b}"] = f["feature"].numpy()
        rec[f"eos_prob.{b}"] = f["eos_prob"].numpy()
        rec[f"alignment.{b}"] = f["alignment"].numpy()
        assert got[b]["feature"].shape == f["feature"].shape
        assert float((got[b]["feature"] - f["feature"]).abs().max()) < 2e-4
        assert torch.equal(got[b]["alignment"], f["alignment"])
    rec["n"] = len(fin)
    # every step sees the same input frame: rows 4k .. 4k+3 of a feature differ only through positions / caches
    np.savez_compressed(os.path.join(OUT, "s2st_tiny_speaker.npz"), **rec)
    print("speaker golden: loss %.5f rows %d lens %s" % (float(loss), rows, [int(f["feature"].shape[0]) for f in fin]))


if __name__ == "__main__":
    main()
