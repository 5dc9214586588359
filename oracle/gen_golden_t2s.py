#!/usr/bin/env python3
"""Golden vectors for the ``t2s_transformer`` variant from the REFERENCE (build container only):
    python oracle/gen_golden_t2s.py        # writes tests/golden/s2st_tiny_t2s.npz
TEST INFRASTRUCTURE: examples/s2s_trans/models/t2s_transformer.py's model through its own ``build_model`` +
examples/s2s_trans/criterions/t2s_loss.py's criterion (forward + backward) on the seeded tiny batch (text side as the
encoder input) with name-keyed synthetic weights."""
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
import gen_golden as GG  # noqa: E402
from configs import CONFIGS, golden_sample  # noqa: E402
from synth_weights import load_synth  # noqa: E402
import s2st_oracle as O  # noqa: E402


def build(a):
    ns = argparse.Namespace(**vars(a))
    t2s_arch(ns)
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
    model = T2STransformerModel.build_model(ns, FakeTask)
    load_synth(model, seed=0)
    return model.train(), FakeTask


def main():
    a = O.make_args(**CONFIGS["tiny_t2s"])
    model, task = build(a)
    crit = T2SCriterion(task, False, a.n_frames_per_step, False, 0.4, a.bce_pos_weight, 0.0)
    sample = dict(golden_sample("tiny", 0), speaker=None)
    out = {}
    loss, ss, log = crit(model, sample)
    for k, v in log.items():
        out[f"log.{k}"] = np.asarray(float(v))
    loss.backward()
    named = dict(model.named_parameters())
    gn = {n: float(p.grad.norm()) for n, p in named.items() if p.grad is not None}
    out["grad_norm_names"] = np.array(sorted(gn))
    out["grad_norms"] = np.array([gn[k] for k in sorted(gn)], dtype=np.float64)
    for n in sorted(gn):
        out[f"gsub.{n}"] = GG.gsub(GG.to_np(named[n].grad))
    sd = model.state_dict()
    out["sd_names"] = np.array(list(sd.keys()))
    out["sd_shapes"] = np.array([",".join(str(int(s)) for s in v.shape) for v in sd.values()])
    for k, v in sd.items():
        if "running_" in k:
            out[f"buf.{k}"] = GG.to_np(v)
    model2, _ = build(a)
    with torch.no_grad():
        post, eos, extra = model2(src_tokens=sample["src_text"], src_lengths=sample["src_text_len"],
                                  prev_output_tokens=sample["net_input"]["prev_output_tokens"], incremental_state=None,
                                  target_lengths=sample["target_lengths"], speaker=None)
        enc = model2.encoder(sample["src_text"], sample["src_text_len"])
    for k, t in dict(post_feat_out=post, eos_out=eos, feature_out=extra["feature_out"], attn=extra["attn"],
                     encoder_out=enc["encoder_out"][0]).items():
        out[f"out.{k}"] = GG.to_np(t).astype(np.float32)
    out["int.stop_idx"] = GG.to_np(O.stop_indices(eos))
    path = os.path.join(GG.ROOT, "tests", "golden", "s2st_tiny_t2s.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), {k: float(v) for k, v in log.items()})


if __name__ == "__main__":
    main()
