#!/usr/bin/env python3
"""Golden vectors for the HuBERT front end from the REFERENCE HubertModel (build container only):
    python oracle/gen_golden_hubert.py      # writes tests/golden/hubert_{tiny,base}.npz
TEST INFRASTRUCTURE: imports /root/reference (with oracle/ref_shims), builds HubertModel through its own
constructor, loads name-keyed synthetic weights and runs extract_features (eval, no mask) on seeded audio.
Stores outputs only (tiny: the whole tensor; base: checksums + a strided sample)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
for _n, _t in dict(float=float, int=int, bool=bool, object=object, complex=complex, str=str).items():
    if not hasattr(np, _n):
        setattr(np, _n, _t)
torch._C.has_cudnn = False

import fairseq  # noqa: E402,F401
from fairseq.data import Dictionary  # noqa: E402
from fairseq.models.hubert.hubert import HubertConfig, HubertModel  # noqa: E402
from fairseq.tasks.hubert_pretraining import HubertPretrainingConfig  # noqa: E402

import hubert_oracle as HO  # noqa: E402
from synth_weights import synth_tensor  # noqa: E402


def build(cfg):
    c = HubertConfig()
    c.conv_feature_layers = repr(cfg["conv"])
    c.encoder_embed_dim, c.encoder_layers = cfg["embed"], cfg["layers"]
    c.encoder_attention_heads, c.encoder_ffn_embed_dim = cfg["heads"], cfg["ffn"]
    c.conv_pos, c.conv_pos_groups, c.label_rate, c.final_dim = cfg["conv_pos"], cfg["conv_pos_groups"], 50, 16
    t = HubertPretrainingConfig()
    t.sample_rate = 16000
    d = Dictionary()
    for i in range(10):
        d.add_symbol(str(i))
    m = HubertModel(c, t, [d])
    sd = m.state_dict()
    m.load_state_dict({k: torch.from_numpy(synth_tensor(k, tuple(v.shape), 0)).to(v.dtype) for k, v in sd.items()})
    return m.eval()


def main():
    out = os.path.join(ROOT, "tests", "golden")
    for name, (B, N, seed) in {"tiny": (3, 4000, 7), "base": (2, 16000, 8)}.items():
        cfg = HO.HUBERT_CONFIGS[name]
        m = build(cfg)
        wave, pad, lens = HO.synth_audio(B, N, seed)
        with torch.no_grad():
            y, fpm = m.extract_features(wave, pad)
            yo, fo = HO.extract_features(HO.synth_state(cfg), cfg, wave, pad)
        err = float((y - yo).abs().max() / y.abs().max())
        assert torch.equal(fpm, fo) and err < 2e-5, (name, err)
        rec = dict(B=B, N=N, seed=seed, frame_pad=fpm.numpy(), sum=np.array([float(y.sum()), float(y.abs().sum()),
                   float((y.double() ** 2).sum().sqrt())]), oracle_rel_err=err)
        if name == "tiny":
            rec["out"] = y.numpy()
        else:
            rec["sample"] = y.numpy()[:, ::5, ::16].copy()
        np.savez_compressed(os.path.join(out, f"hubert_{name}.npz"), **rec)
        print(name, tuple(y.shape), "oracle vs reference rel err", err)


if __name__ == "__main__":
    main()
