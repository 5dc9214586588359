#!/usr/bin/env python3
"""Golden vectors of the ST / ASR pre-training stage from the REFERENCE classes (build container only):
    python oracle/gen_golden_s2t.py        # writes tests/golden/s2t_tiny.npz
TEST INFRASTRUCTURE.  Builds the reference's `S2TTransformerModel` (fairseq/models/speech_to_text/s2t_transformer_me.py:
82-330, arch s2t_transformer_hubert, no HuBERT: fbank input) and `LabelSmoothedCrossEntropyCriterion`
(examples/s2s_trans/criterions/s2t_loss.py:57-198) through their own constructors, loads name-keyed synthetic weights,
and records, for --test-type asr AND st on the tiny golden batch: logging output, logits, encoder output, every gradient
tensor (sampled), the state_dict contract, and three updates through the reference's Adam / clip_grad_norm_."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.argv = [sys.argv[0]]
import gen_golden as GG  # noqa: E402  (reference import path + stand-ins)
import fairseq.models as _fm  # noqa: E402
import fairseq.criterions as _fc  # noqa: E402
# (gen_golden imported this package, whose plugin registered the names first: free them for the reference's own modules)
for _reg in (_fm.MODEL_REGISTRY, _fm.ARCH_MODEL_REGISTRY, _fm.ARCH_MODEL_NAME_REGISTRY, _fm.ARCH_CONFIG_REGISTRY):
    for _k in [k for k in _reg if str(k).startswith("s2t_transformer_hubert")]:
        _reg.pop(_k, None)
_fm.ARCH_MODEL_INV_REGISTRY.pop("s2t_transformer_hubert", None)
_fc.CRITERION_REGISTRY.pop("s2t_loss", None)
getattr(_fc, "CRITERION_CLASS_NAMES", set()).discard("LabelSmoothedCrossEntropyCriterion")
from fairseq.models.speech_to_text.s2t_transformer_me import S2TTransformerModel, base_architecture  # noqa: E402
from examples.s2s_trans.criterions.s2t_loss import LabelSmoothedCrossEntropyCriterion  # noqa: E402

import s2st_oracle as O  # noqa: E402
import s2t_oracle as SO  # noqa: E402
from configs import S2T_TINY, golden_sample  # noqa: E402
from synth_weights import load_synth  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def build_reference(cfg, test_type):
    a = SO.make_args(**cfg)
    ns = argparse.Namespace(
        encoder_layers=a.encoder_transformer_layers, decoder_layers=a.decoder_transformer_layers,
        encoder_embed_dim=a.encoder_embed_dim, decoder_embed_dim=a.decoder_embed_dim,
        encoder_ffn_embed_dim=a.encoder_ffn_embed_dim, decoder_ffn_embed_dim=a.decoder_ffn_embed_dim,
        encoder_attention_heads=a.encoder_attention_heads, decoder_attention_heads=a.decoder_attention_heads,
        encoder_normalize_before=a.encoder_normalize_before, decoder_normalize_before=a.decoder_normalize_before,
        dropout=a.dropout, attention_dropout=a.attention_dropout, activation_dropout=a.activation_dropout,
        conv_kernel_sizes=a.conv_kernel_sizes, input_feat_per_channel=80, input_channels=1, hubert_hidden=768,
        use_hubert="false", max_source_positions=3000, max_target_positions=2400, no_scale_embedding=False,
        load_pretrained_encoder_from=None, load_pretrained_hubert_from=None)
    base_architecture(ns)
    src_d, tgt_d = GG.make_dict(a.src_vocab_size), GG.make_dict(a.tgt_vocab_size)

    class FakeTask:
        source_dictionary, target_dictionary, src_dict, tgt_dict, args = src_d, tgt_d, src_d, tgt_d, ns

    model = S2TTransformerModel.build_model(ns, FakeTask)
    crit = LabelSmoothedCrossEntropyCriterion(FakeTask, False, a.label_smoothing, 0, True, test_type)
    return a, model, crit


def main():
    cfg = S2T_TINY
    sample = golden_sample("tiny", 0)
    ni = sample["net_input"]
    ni["collated_audios_orig"], ni["padding_mask"] = None, None
    rec = {}
    for tt in ("asr", "st"):
        a, model, crit = build_reference(cfg, tt)
        load_synth(model, 0)
        model.train()
        loss, ss, log = crit(model, sample)
        loss.backward()
        for k, v in log.items():
            rec[f"{tt}.log.{k}"] = np.asarray(float(v))
        rec[f"{tt}.sample_size"] = np.asarray(float(ss))
        key = "src" if tt == "asr" else "tgt"
        with torch.no_grad():
            logits, _ = model(ni["src_speech"], ni["src_speech_lens"], None, None, ni[f"prev_{key}_text_tokens"])
            enc = model.encoder(ni["src_speech"], ni["src_speech_lens"], None, None)
        rec[f"{tt}.logits"] = GG.to_np(logits)
        rec[f"{tt}.encoder_out"] = GG.sub(GG.to_np(enc["encoder_out"][0]))
        named = dict(model.named_parameters())
        for n in sorted(named):
            if named[n].grad is not None:
                rec[f"{tt}.gsub.{n}"] = GG.gsub(GG.to_np(named[n].grad))
        rec[f"{tt}.grad_none"] = np.array([n for n, p in named.items() if p.grad is None])
        if tt == "asr":
            sd = model.state_dict()
            rec["sd_names"] = np.array(list(sd.keys()))
            rec["sd_shapes"] = np.array([",".join(str(int(s)) for s in v.shape) for v in sd.values()])
        # the oracle must reproduce it
        m = SO.S2TModel(a)
        load_synth(m, 0)
        m.train()
        l2, _, log2, outs = SO.criterion_forward(m, sample, tt, a.label_smoothing)
        l2.backward()
        assert abs(float(l2) - float(loss)) < 2e-5 * abs(float(loss)), (tt, float(l2), float(loss))
        assert log2["n_correct"] == int(log["n_correct"]) and log2["total"] == int(log["total"])
        assert float((outs["logits"] - logits).abs().max()) < 2e-4
        mine = dict(m.named_parameters())
        assert set(mine) == set(named), set(mine) ^ set(named)
        gmax = max(float(p.grad.norm()) for p in named.values() if p.grad is not None)
        for n, p in named.items():
            if p.grad is not None:  # (floor: key-projection biases have a mathematically zero gradient)
                d = float((mine[n].grad - p.grad).norm()) / (float(p.grad.norm()) + 1e-3 * gmax)
                assert d < 2e-3, (tt, n, d)
        print(f"[{tt}] loss {float(loss):.6f} nll {float(log['nll_loss']):.6f} ntokens {int(log['ntokens'])} "
              f"acc {int(log['n_correct'])}/{int(log['total'])}; oracle agrees")
    # ---- three updates (test-type st) with the reference's own Adam / clip (trainer.py:838-873) -------------
    a, model, crit = build_reference(cfg, "st")
    load_synth(model, 0)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = GG.RefAdam(params, lr=0.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)  # (fairseq's --adam-betas default)
    lr0, warm, clip = 1e-3, 2, 1.0
    losses, gnorms = [], []
    batches = [golden_sample("tiny", 0), golden_sample("tiny", 1), golden_sample("tiny", 0)]
    for u, s in enumerate(batches):
        s["net_input"]["collated_audios_orig"], s["net_input"]["padding_mask"] = None, None
        lr = float(O.inverse_sqrt_lr(u, lr0, warm))
        for g in opt.param_groups:
            g["lr"] = lr
        opt.zero_grad()
        loss, ss, log = crit(model, s)
        loss.backward()
        for p in params:
            if p.grad is not None:
                p.grad.mul_(1.0 / float(ss))
        gn = GG.ref_clip(params, clip)
        opt.step()
        losses.append(float(loss))
        gnorms.append(float(gn))
    rec["train.loss"], rec["train.gnorm"] = np.asarray(losses), np.asarray(gnorms)
    rec["train.hparams"] = np.asarray([lr0, warm, clip])
    named = dict(model.named_parameters())
    rec["train.param_norm_names"] = np.array(sorted(named))
    rec["train.param_norms"] = np.asarray([float(named[n].norm()) for n in sorted(named)])
    for n in ("encoder.transformer_layers.0.self_attn.q_proj.weight", "decoder.layers.0.encoder_attn.k_proj.weight",
              "decoder.embed_tokens.weight", "decoder.output_projection.weight", "encoder.layer_norm.weight"):
        rec[f"train.param.{n}"] = GG.sub(GG.to_np(named[n]))
    np.savez_compressed(os.path.join(OUT, "s2t_tiny.npz"), **rec)
    print("s2t golden ok: train losses", losses, "gnorms", gnorms)


if __name__ == "__main__":
    main()
