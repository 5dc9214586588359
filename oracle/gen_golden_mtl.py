#!/usr/bin/env python3
"""Golden vectors for the ``s2st_transformer_mtl`` variant from the REFERENCE (build container only):
    python oracle/gen_golden_mtl.py        # writes tests/golden/s2st_tiny_mtl.npz
TEST INFRASTRUCTURE: builds examples/s2s_trans/models/s2st_transformer_mtl.py's model through its own ``build_model``
and runs examples/s2s_trans/criterions/s2st_loss_mtl.py's criterion (forward + backward) on the seeded tiny batch with
name-keyed synthetic weights.  Stores losses, outputs, gradient norms + samples, state-dict names."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# the reference's mtl plugin has to be imported BEFORE this repo's package (gen_golden imports it for the synthetic
# corpus): both register "s2st_transformer_mtl" with fairseq and the second registration is refused
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
for _n, _t in dict(float=float, int=int, bool=bool, object=object, complex=complex, str=str).items():
    if not hasattr(np, _n):
        setattr(np, _n, _t)
torch._C.has_cudnn = False
import fairseq  # noqa: E402,F401
from examples.s2s_trans.models.s2st_transformer_mtl import S2STTransformerModel as MTLModel, base_architecture as mtl_arch  # noqa: E402
from examples.s2s_trans.criterions.s2st_loss_mtl import Tacotron2Criterion as MTLCriterion  # noqa: E402
import gen_golden as GG  # noqa: E402
from configs import CONFIGS, golden_sample  # noqa: E402
from synth_weights import load_synth  # noqa: E402
import s2st_oracle as O  # noqa: E402



def main():
    cfg = CONFIGS["tiny_mtl"]
    a = O.make_args(**cfg)
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
    model.train()
    crit = MTLCriterion(FakeTask, False, a.n_frames_per_step, False, 0.4, a.bce_pos_weight, a.ctc_weight, a.ctc_weight_tgt)
    sample = golden_sample("tiny", 0)
    sample = dict(sample, speaker=None)
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
    # tensors of the same forward (fresh model: single BatchNorm update)
    model2 = MTLModel.build_model(ns, FakeTask)
    load_synth(model2, seed=0)
    model2.train()
    ni = sample["net_input"]
    with torch.no_grad():
        post, eos, extra = model2(src_tokens=ni["src_speech"], src_lengths=ni["src_speech_lens"],
                                  prev_output_tokens=ni["prev_output_tokens"], incremental_state=None,
                                  target_lengths=sample["target_lengths"], speaker=None)
        lp_tgt = model2.get_normalized_probs((post, eos, extra["out_middle_layers_decoder"]), True, tag="ctc_tgt")
    for k, t in dict(post_feat_out=post, eos_out=eos, feature_out=extra["feature_out"], ctc_tgt_lprobs=lp_tgt).items():
        out[f"out.{k}"] = GG.to_np(t).astype(np.float32)
    out["int.ctc_tgt_greedy"] = GG.to_np(O.ctc_greedy_path(lp_tgt.transpose(0, 1), sample["target_lengths"]))
    out["int.stop_idx"] = GG.to_np(O.stop_indices(eos))
    path = os.path.join(GG.ROOT, "tests", "golden", "s2st_tiny_mtl.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), {k: float(v) for k, v in log.items()})


if __name__ == "__main__":
    main()
