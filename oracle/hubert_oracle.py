"""CPU restatement (torch fp32) of the frozen HuBERT front end of config 4 -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product path never does.  Follows (paths under /root/reference):
  fairseq/models/hubert/hubert.py:412-461 (forward with features_only), :400-410
  (forward_padding_mask), :518-534 (extract_features);
  fairseq/models/wav2vec/wav2vec2.py:736-814 (ConvFeatureExtractionModel, mode "default": conv0 +
  GroupNorm(C, C) + GELU, then conv + GELU, no conv bias), :817-905 (TransformerEncoder: zero the
  padded frames, weight-normed grouped pos_conv + SamePad + GELU, residual, LayerNorm, post-LN layers),
  :908-1020 (TransformerSentenceEncoderLayer, layer_norm_first=False), eval mode (no dropout,
  no layerdrop, mask=False).
Pinned by tests/golden/hubert_*.npz, generated from the reference's HubertModel itself with
name-keyed synthetic weights (oracle/gen_golden_hubert.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from synth_weights import synth_tensor

TINY = dict(conv=[(32, 10, 5)] + [(32, 3, 2)] * 4 + [(32, 2, 2)] * 2, embed=64, layers=2, heads=4, ffn=128,
            conv_pos=16, conv_pos_groups=4)
BASE = dict(conv=[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2, embed=768, layers=12, heads=12,
            ffn=3072, conv_pos=128, conv_pos_groups=16)
HUBERT_CONFIGS = {"tiny": TINY, "base": BASE}


def state_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """Names / shapes of the reference HubertModel.state_dict() entries the forward reads."""
    s: Dict[str, Tuple[int, ...]] = {}
    cin = 1
    for i, (c, k, _) in enumerate(cfg["conv"]):
        s[f"feature_extractor.conv_layers.{i}.0.weight"] = (c, cin, k)
        if i == 0:
            s["feature_extractor.conv_layers.0.2.weight"] = (c,)
            s["feature_extractor.conv_layers.0.2.bias"] = (c,)
        cin = c
    E = cfg["embed"]
    s["layer_norm.weight"] = (cin,)
    s["layer_norm.bias"] = (cin,)
    s["post_extract_proj.weight"] = (E, cin)
    s["post_extract_proj.bias"] = (E,)
    s["encoder.pos_conv.0.bias"] = (E,)
    s["encoder.pos_conv.0.weight_g"] = (1, 1, cfg["conv_pos"])
    s["encoder.pos_conv.0.weight_v"] = (E, E // cfg["conv_pos_groups"], cfg["conv_pos"])
    for l in range(cfg["layers"]):
        p = f"encoder.layers.{l}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (E, E)
            s[p + f"self_attn.{n}.bias"] = (E,)
        s[p + "self_attn_layer_norm.weight"] = (E,)
        s[p + "self_attn_layer_norm.bias"] = (E,)
        s[p + "fc1.weight"] = (cfg["ffn"], E)
        s[p + "fc1.bias"] = (cfg["ffn"],)
        s[p + "fc2.weight"] = (E, cfg["ffn"])
        s[p + "fc2.bias"] = (E,)
        s[p + "final_layer_norm.weight"] = (E,)
        s[p + "final_layer_norm.bias"] = (E,)
    s["encoder.layer_norm.weight"] = (E,)
    s["encoder.layer_norm.bias"] = (E,)
    return s


def synth_state(cfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(synth_tensor(k, v, seed)) for k, v in state_shapes(cfg).items()}


def conv_out_len(n: int, cfg) -> int:
    for _, k, s in cfg["conv"]:
        n = (n - k) // s + 1
    return n


def frame_padding_mask(pad_mask: torch.Tensor, n_frames: int) -> torch.Tensor:
    """hubert.py:400-410: a frame is padding iff ALL samples of its chunk are padding."""
    extra = pad_mask.size(1) % n_frames
    if extra > 0:
        pad_mask = pad_mask[:, :-extra]
    return pad_mask.view(pad_mask.size(0), n_frames, -1).all(-1)


def mha(x, sd, p, heads, key_pad):
    """fairseq MultiheadAttention, self-attention, eval mode (multihead_attention.py:160-385)."""
    T, B, E = x.shape
    dh = E // heads
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]) * dh ** -0.5
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    sh = lambda t: t.contiguous().view(T, B * heads, dh).transpose(0, 1)
    q, k, v = sh(q), sh(k), sh(v)
    w = torch.bmm(q, k.transpose(1, 2)).view(B, heads, T, T)
    if key_pad is not None:
        w = w.masked_fill(key_pad[:, None, None, :], float("-inf"))
    w = torch.softmax(w.float(), dim=-1).view(B * heads, T, T)
    o = torch.bmm(w, v).transpose(0, 1).contiguous().view(T, B, E)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def extract_features(sd: Dict[str, torch.Tensor], cfg, wave: torch.Tensor, pad_mask: torch.Tensor):
    """wave [B, N] fp32, pad_mask [B, N] bool -> (x [B, T', E], frame pad mask [B, T'])."""
    x = wave.unsqueeze(1)
    for i, (c, k, s) in enumerate(cfg["conv"]):  # wav2vec2.py:806-814
        x = F.conv1d(x, sd[f"feature_extractor.conv_layers.{i}.0.weight"], None, stride=s)
        if i == 0:
            x = F.group_norm(x.float(), c, sd["feature_extractor.conv_layers.0.2.weight"],
                             sd["feature_extractor.conv_layers.0.2.bias"], 1e-5)
        x = F.gelu(x)
    x = x.transpose(1, 2)  # hubert.py:426-427
    x = F.layer_norm(x, (x.size(-1),), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)
    fpm = frame_padding_mask(pad_mask, x.size(1))
    x = F.linear(x, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
    # TransformerEncoder.extract_features (wav2vec2.py:868-905)
    x = x.masked_fill(fpm.unsqueeze(-1), 0.0)
    g, v = sd["encoder.pos_conv.0.weight_g"], sd["encoder.pos_conv.0.weight_v"]
    w = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()  # weight_norm(dim=2)
    kp = cfg["conv_pos"]
    xc = F.conv1d(x.transpose(1, 2), w, sd["encoder.pos_conv.0.bias"], padding=kp // 2, groups=cfg["conv_pos_groups"])
    if kp % 2 == 0:
        xc = xc[:, :, :-1]  # SamePad
    x = x + F.gelu(xc).transpose(1, 2)
    E = cfg["embed"]
    x = F.layer_norm(x, (E,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], 1e-5)
    x = x.transpose(0, 1)
    kp_mask = fpm if bool(fpm.any()) else None
    for l in range(cfg["layers"]):
        p = f"encoder.layers.{l}."
        r = x
        x = r + mha(x, sd, p + "self_attn.", cfg["heads"], kp_mask)
        x = F.layer_norm(x, (E,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], 1e-5)
        r = x
        x = F.linear(F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"])), sd[p + "fc2.weight"], sd[p + "fc2.bias"])
        x = F.layer_norm(r + x, (E,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5)
    return x.transpose(0, 1), fpm


def synth_audio(B: int, N: int, seed: int):
    """Seeded waveform batch: N(0, 0.1) samples, right-padded with zeros; lengths sorted descending."""
    g = torch.Generator().manual_seed(seed)
    lens = sorted([N] + [int(N * (0.55 + 0.4 * float(torch.rand(1, generator=g)))) for _ in range(B - 1)], reverse=True)
    wave = 0.1 * torch.randn(B, N, generator=g)
    pad = torch.zeros(B, N, dtype=torch.bool)
    for b, n in enumerate(lens):
        wave[b, n:] = 0
        pad[b, n:] = True
    return wave, pad, torch.tensor(lens)
