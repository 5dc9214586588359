#!/usr/bin/env python3
"""Golden vectors for BASELINE.json configs[3]: --use-hubert (frozen hubert_base front end) + base model + aux
ASR/ST decoders as ONE training step, from the REFERENCE itself (build container only):

    python oracle/gen_golden_hubert_train.py      # writes tests/golden/s2st_hubert_train.npz

TEST INFRASTRUCTURE.  The reference builds its HuBERT through ``load_model_ensemble_and_task`` from
``hubert_base_ls960.pt`` (s2st_transformer.py:685-703), which is not on disk (and needs a functional OmegaConf):
``build_hubert`` is therefore replaced by a function returning the reference's own ``HubertModel`` built directly
with name-keyed synthetic weights (as oracle/gen_golden_hubert.py does).  Everything downstream -- the encoder's
HuBERT branch (:245-252), model, ``Tacotron2Criterion`` (incl. the CTC-length quirk of SURVEY B.7), the reference's
Adam / clip -- is the reference's code.  Stores losses, output checksums, gradient norms + samples of every gradient
tensor and two optimizer updates.
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as GG  # noqa: E402  (sets up sys.path / preludes, imports the reference)
import gen_golden_hubert as GH  # noqa: E402
import hubert_oracle as HO  # noqa: E402
from configs import CONFIGS, hubert_train_sample  # noqa: E402
from synth_weights import load_synth  # noqa: E402
import s2st_oracle as O  # noqa: E402

from examples.s2s_trans.models.s2st_transformer import S2STTransformerModel  # noqa: E402


def build(cfg, hub):
    S2STTransformerModel.build_hubert = classmethod(lambda cls, args: hub)
    a, model, crit = GG.build_reference(cfg)
    # the synthetic-weight loader addresses the s2st parameters; HuBERT keeps the weights GH.build gave it
    hub_sd = {k: v.clone() for k, v in hub.state_dict().items()}
    load_synth(model, seed=0, skip_prefix="encoder.hubert.")
    hub.load_state_dict(hub_sd)
    return a, model, crit


def main(which):
    out_dir = os.path.join(GG.ROOT, "tests", "golden")
    name = "hubert_train"
    cfg = CONFIGS[name]
    geo = HO.HUBERT_CONFIGS[cfg.get("hubert_geometry_name", "base")]
    out = {}
    sample = hubert_train_sample(0)
    hub = GH.build(geo)
    a, model, crit = build(cfg, hub)
    model.train()
    try:
        loss, ss, log = crit(model, sample)
    except Exception as e:  # fbank-derived CTC lengths against HuBERT-rate encoder frames
        print("reference raised:", type(e).__name__, e)
        raise
    for k, v in log.items():
        out[f"log.{k}"] = np.asarray(float(v))
    loss.backward()
    named = dict(model.named_parameters())
    gn = {n: float(p.grad.norm()) for n, p in named.items() if p.grad is not None}
    assert not any(n.startswith("encoder.hubert.") for n in gn), "the front end is frozen"
    out["grad_norm_names"] = np.array(sorted(gn))
    out["grad_norms"] = np.array([gn[k] for k in sorted(gn)], dtype=np.float64)
    for n in sorted(gn):
        out[f"gsub.{n}"] = GG.gsub(GG.to_np(named[n].grad))
    # tensors of the same forward (fresh model: single BatchNorm update)
    hub2 = GH.build(geo)
    a2, model2, crit2 = build(cfg, hub2)
    model2.train()
    ni = sample["net_input"]
    with torch.no_grad():
        net = model2(src_tokens=ni["src_speech"], src_lengths=ni["src_speech_lens"],
                     collated_audios=ni["collated_audios_orig"], padding_mask=ni["padding_mask"],
                     prev_output_tokens=ni["prev_output_tokens"], prev_src_text_tokens=ni["prev_src_text_tokens"],
                     prev_tgt_text_tokens=ni["prev_tgt_text_tokens"], incremental_state=None,
                     target_lengths=sample["target_lengths"], speaker=None)
        (post, eos, extra), asr, st = net
        feats, fpm = hub2.extract_features(ni["collated_audios_orig"], ni["padding_mask"])
    out["int.hubert_frames"] = GG.to_np((~fpm).long().sum(-1))
    out["int.stop_idx"] = GG.to_np(O.stop_indices(eos))
    tens = {"post_feat_out": post, "eos_out": eos, "feature_out": extra["feature_out"], "asr_logits": asr[0],
            "st_logits": st[0], "hubert_features": feats}
    for k, t in tens.items():
        t = GG.to_np(t).astype(np.float64)
        out[f"sum.{k}"] = np.array([t.sum(), np.abs(t).sum(), float(np.sqrt((t ** 2).sum()))])
    sd = model2.state_dict()
    for k, v in sd.items():
        if "running_" in k and "hubert" not in k:
            out[f"buf.{k}"] = GG.to_np(v)
    # two optimizer updates with the reference's Adam / clip over the trainable parameters
    hub3 = GH.build(geo)
    a3, model3, crit3 = build(cfg, hub3)
    model3.train()
    params = [p for n, p in model3.named_parameters() if not n.startswith("encoder.hubert.")]
    opt = GG.RefAdam(params, lr=0.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    LR, WARM, CLIP = 1e-3, 2, 0.02
    losses, gnorms = [], []
    for u in range(2):
        s = hubert_train_sample(u % 2)
        for p in params:
            p.grad = None
        loss, ss, log = crit3(model3, s)
        loss.backward()
        for p in params:
            if p.grad is not None:
                p.grad.mul_(1.0 / float(ss))
        gnorm = GG.ref_clip(params, CLIP)
        for g in opt.param_groups:
            g["lr"] = O.inverse_sqrt_lr(u, LR, WARM)
        opt.step()
        losses.append(float(loss))
        gnorms.append(float(gnorm))
    out["train.loss"], out["train.gnorm"] = np.array(losses), np.array(gnorms)
    out["train.hparams"] = np.array([LR, WARM, CLIP, 2])
    pn = {n: float(p.detach().norm()) for n, p in model3.named_parameters() if not n.startswith("encoder.hubert.")}
    out["train.param_norm_names"] = np.array(sorted(pn))
    out["train.param_norms"] = np.array([pn[k] for k in sorted(pn)], dtype=np.float64)
    path = os.path.join(out_dir, f"s2st_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)/1e6:.2f} MB, loss={losses}, gnorm={gnorms}, log={ {k: float(v) for k, v in log.items()} }")


if __name__ == "__main__":
    main(sys.argv[1:])
