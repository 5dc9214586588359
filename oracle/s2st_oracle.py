"""CPU oracle for the s2st_transformer training path.

TEST INFRASTRUCTURE ONLY.  A plain PyTorch fp32 (CPU) restatement of the reference
algorithm; it is imported only by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never by the product path, which runs the HIP
kernels behind ``include/s2st_hip.h``.

Parity status: PINNED against outputs of the reference itself (imported in the build
container by ``oracle/gen_golden.py`` with the stubs in ``oracle/ref_shims``); the
resulting vectors are committed under ``tests/golden/`` and checked by
``tests/test_oracle_golden.py``.  The reference ships no tests for this path
(SURVEY.md section 4), and the arithmetic of ``F.linear / conv1d / layer_norm / batch_norm``
is PyTorch's (reference pins no version; goldens were generated on torch 2.10 CPU).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  No fairseq import, no reference source.
"""
from __future__ import annotations

import argparse
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

PAD = 1  # fairseq Dictionary.pad()


# ----------------------------------------------------------------------------------------
# arguments
# ----------------------------------------------------------------------------------------
def make_args(**kw) -> argparse.Namespace:
    """Namespace with the defaults of ``base_architecture``
    (examples/s2s_trans/models/s2st_transformer.py:792-830) plus the task / criterion
    flags the constructors read."""
    a = argparse.Namespace(**kw)

    def d(name, val):
        if not hasattr(a, name):
            setattr(a, name, val)

    d("dropout", 0.1)
    d("output_frame_dim", 80)
    d("middle_layers", "6")
    d("conv_kernel_sizes", "5,5")
    # reference quirk: --conv-channels is ignored (typo `conv_chaFnnels`, :802)
    a.conv_channels = 1024
    d("encoder_transformer_layers", 12)
    d("encoder_embed_dim", 512)
    d("encoder_ffn_embed_dim", 4 * a.encoder_embed_dim)
    d("encoder_normalize_before", True)
    d("encoder_attention_heads", 4)
    d("attention_dropout", a.dropout)
    d("activation_dropout", a.dropout)
    d("activation_fn", "relu")
    d("prenet_dropout", 0.5)
    d("prenet_layers", 2)
    d("prenet_dim", 256)
    d("postnet_dropout", 0.5)
    d("postnet_layers", 5)
    d("postnet_conv_dim", 512)
    d("postnet_conv_kernel_size", 5)
    d("asr_decoder_layers", 6)
    d("st_decoder_layers", 6)
    d("asr_decoder_embed_dim", 256)
    d("st_decoder_embed_dim", 256)
    d("decoder_transformer_layers", 6)
    d("decoder_embed_dim", 512)
    d("decoder_ffn_embed_dim", 4 * a.decoder_embed_dim)
    d("decoder_normalize_before", False)
    d("decoder_attention_heads", 4)
    # task / criterion side
    d("n_frames_per_step", 4)
    d("input_feat_per_channel", 80)
    d("input_channels", 1)
    d("max_source_positions", 3000)
    d("max_target_positions", 2400)
    d("no_scale_embedding", False)
    d("hubert_hidden", 768)
    d("use_hubert", "false")
    d("ctc_weight", 0.0)
    d("asr_ce_weight", 0.0)
    d("st_ce_weight", 0.0)
    d("src_vocab_size", 44)
    d("tgt_vocab_size", 74)
    d("bce_pos_weight", 5.0)
    d("label_smoothing", 0.1)
    d("l1_loss_weight", 1.0)
    d("mse_loss_weight", 1.0)
    d("eos_loss_weight", 1.0)
    d("attn_loss_weight", 1.0)
    d("use_guided_attention_loss", False)
    d("guided_attention_loss_sigma", 0.4)
    d("speaker_embed_dim", 64)  # base_architecture (:796-797)
    d("speaker_embed_dim_dec", 64)
    return a


# ----------------------------------------------------------------------------------------
# dropout sites
# ----------------------------------------------------------------------------------------
# Every dropout of the path goes through drop().  By default it IS F.dropout (torch's generator, as in the reference:
# fairseq/modules/fairseq_dropout.py:16-27).  Inside ``with injected_masks(provider):`` the keep decision comes from
# ``provider(site, layout, x, p)`` instead -- a 0/1 tensor of x's shape -- and the kept elements are scaled by 1 / (1 - p)
# exactly as F.dropout scales them: the way tests/test_dropout_parity.py hands the oracle the masks the HIP engine used
# (it keeps none; a mask is a function of (site seed, element index)), so that a misplaced site, a missing 1 / (1 - p) or
# a forward / backward mask mismatch in the fused kernels shows against the reference's placement below.
# A site's name = where it sits + kind + ordinal there (the engine's log uses the same scheme, include/s2st_hip.h):
#   "<ctx>/attn<k>" probabilities [B*H, T, S]       (multihead_attention.py:360-366)
#   "<ctx>/lin<k>"  outputs of linear layers        (transformer_layer.py:150-162, 384-431; tacotron2.py:95-98)
#   "<ctx>/rows0"   after the position add          (s2st_transformer.py:197-208, 385-388; transformer_decoder.py:281-326)
#   "<ctx>/norm<k>" after conv -> BatchNorm (-> tanh) (tacotron2.py:122-126; t2s_transformer.py:55-66)
# layout says how x's axes relate to the engine's [B][T][C] rows: "tbc", "btc", "bct", or "attn".
_MASK_PROVIDER = None


class injected_masks:
    def __init__(self, provider):
        self.provider = provider

    def __enter__(self):
        global _MASK_PROVIDER
        self.prev, _MASK_PROVIDER = _MASK_PROVIDER, self.provider
        return self

    def __exit__(self, *exc):
        global _MASK_PROVIDER
        _MASK_PROVIDER = self.prev


def drop(x, p, training, site, layout):
    if _MASK_PROVIDER is None or not training or p <= 0.0:
        return F.dropout(x, p=p, training=training)
    keep = _MASK_PROVIDER(site, layout, x, p)
    assert keep.shape == x.shape, (site, tuple(keep.shape), tuple(x.shape))
    return x * keep.to(x.dtype) * (1.0 / (1.0 - p))


def name_sites(model):
    """Give every module that owns a dropout site its place name (ctx): encoder / decoder / aux-decoder layers by index."""
    def layers(mods, pre):
        for i, l in enumerate(mods):
            l.site = f"{pre}.L{i}"
            l.self_attn.site, l.self_attn.site_k = l.site, 0
            if hasattr(l, "encoder_attn"):
                l.encoder_attn.site, l.encoder_attn.site_k = l.site, 1
    layers(model.encoder.transformer_layers, "enc")
    if hasattr(model, "decoder") and hasattr(model.decoder, "transformer_layers"):
        layers(model.decoder.transformer_layers, "dec")
    for who, dec in (("asr", getattr(model, "aux_asr_decoder", None)), ("st", getattr(model, "aux_st_decoder", None))):
        if dec is not None:
            dec.site = who
            layers(dec.layers, who)
    return model


# ----------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------
def lengths_to_padding_mask(lens: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    """True at padded positions (fairseq/data/data_utils.py:532-536)."""
    mx = int(lens.max().item()) if max_len is None else max_len
    return torch.arange(mx, device=lens.device).unsqueeze(0) >= lens.unsqueeze(1)


def sinusoidal_table(num: int, dim: int, padding_idx: Optional[int]) -> torch.Tensor:
    """tensor2tensor layout [sin | cos], exponent log(1e4)/(half-1)
    (fairseq/modules/sinusoidal_positional_embedding.py:35-58)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(num, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    t = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(num, -1)
    if dim % 2 == 1:
        t = torch.cat([t, torch.zeros(num, 1)], dim=1)
    if padding_idx is not None:
        t[padding_idx, :] = 0
    return t


def make_positions(x: torch.Tensor, padding_idx: int) -> torch.Tensor:
    """Non-pad symbols numbered from padding_idx+1, pads keep padding_idx
    (fairseq/utils.py:254-264).  Also called on a *bool padding mask* by the speech
    encoder/decoder: True.ne(1) is False, so padded frames get index 1 (= zero row)."""
    m = x.ne(padding_idx).int()
    return (torch.cumsum(m, dim=1).type_as(m) * m).long() + padding_idx


def positional_embedding(x: torch.Tensor, dim: int, padding_idx: int = PAD) -> torch.Tensor:
    """[B, T] tokens-or-mask -> [B, T, dim] (sinusoidal_positional_embedding.py:60-105)."""
    b, t = x.shape
    tab = sinusoidal_table(padding_idx + 1 + t, dim, padding_idx)
    return tab.index_select(0, make_positions(x, padding_idx).view(-1)).view(b, t, dim)


class _PositionalEmbeddingState(nn.Module):
    """Holds the `_float_tensor` buffer SinusoidalPositionalEmbedding registers
    (sinusoidal_positional_embedding.py:29) so state_dict keys match Appendix A."""

    def __init__(self):
        super().__init__()
        self.register_buffer("_float_tensor", torch.zeros(1))


class MultiheadAttention(nn.Module):
    """q,k,v,out projections with bias; q scaled by head_dim**-0.5 before QK^T; additive
    -inf masks; softmax in fp32; dropout on probabilities
    (fairseq/modules/multihead_attention.py:194-385; the fused fast path :160-192 computes
    the same function)."""

    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0.0):
        super().__init__()
        kdim = embed_dim if kdim is None else kdim
        vdim = embed_dim if vdim is None else vdim
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
        self.dropout = dropout
        self.site, self.site_k = "", 0  # (name_sites)
        self.k_proj = nn.Linear(kdim, embed_dim)
        self.v_proj = nn.Linear(vdim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)

    def forward(self, query, key, value, key_padding_mask=None, attn_mask=None,
                need_head_weights=False):
        # query [T, B, C], key/value [S, B, Ck]
        T, B, C = query.shape
        S = key.shape[0]
        H, Dh = self.num_heads, self.head_dim
        q = self.q_proj(query) * self.scaling
        k = self.k_proj(key)
        v = self.v_proj(value)
        q = q.contiguous().view(T, B * H, Dh).transpose(0, 1)
        k = k.contiguous().view(S, B * H, Dh).transpose(0, 1)
        v = v.contiguous().view(S, B * H, Dh).transpose(0, 1)
        w = torch.bmm(q, k.transpose(1, 2))  # [B*H, T, S]
        if attn_mask is not None:
            w = w + attn_mask.unsqueeze(0)
        if key_padding_mask is not None:
            w = w.view(B, H, T, S).masked_fill(
                key_padding_mask.unsqueeze(1).unsqueeze(2).to(torch.bool), float("-inf")
            ).view(B * H, T, S)
        p = F.softmax(w.float(), dim=-1).type_as(w)
        pd = drop(p, self.dropout, self.training, f"{self.site}/attn{self.site_k}", "attn")
        o = torch.bmm(pd, v)  # [B*H, T, Dh]
        o = o.transpose(0, 1).contiguous().view(T, B, C)
        o = self.out_proj(o)
        hw = p.view(B, H, T, S).transpose(1, 0) if need_head_weights else None  # [H,B,T,S]
        return o, hw


class TransformerEncoderLayer(nn.Module):
    """fairseq/modules/transformer_layer.py:107-165."""

    def __init__(self, dim, heads, ffn, normalize_before, dropout, attn_dropout, act_dropout):
        super().__init__()
        self.self_attn = MultiheadAttention(dim, heads, dropout=attn_dropout)
        self.self_attn_layer_norm = nn.LayerNorm(dim)
        self.fc1 = nn.Linear(dim, ffn)
        self.fc2 = nn.Linear(ffn, dim)
        self.final_layer_norm = nn.LayerNorm(dim)
        self.normalize_before = normalize_before
        self.p, self.pa = dropout, act_dropout
        self.site = ""

    def forward(self, x, pad_mask):
        r = x
        if self.normalize_before:
            x = self.self_attn_layer_norm(x)
        x, _ = self.self_attn(x, x, x, key_padding_mask=pad_mask)
        x = r + drop(x, self.p, self.training, f"{self.site}/lin0", "tbc")
        if not self.normalize_before:
            x = self.self_attn_layer_norm(x)
        r = x
        if self.normalize_before:
            x = self.final_layer_norm(x)
        x = drop(F.relu(self.fc1(x)), self.pa, self.training, f"{self.site}/lin1", "tbc")
        x = r + drop(self.fc2(x), self.p, self.training, f"{self.site}/lin2", "tbc")
        if not self.normalize_before:
            x = self.final_layer_norm(x)
        return x


class TransformerDecoderLayer(nn.Module):
    """fairseq/modules/transformer_layer.py:301-446 (no incremental state)."""

    def __init__(self, dim, heads, ffn, enc_dim, normalize_before, dropout, attn_dropout,
                 act_dropout):
        super().__init__()
        self.self_attn = MultiheadAttention(dim, heads, dropout=attn_dropout)
        self.self_attn_layer_norm = nn.LayerNorm(dim)
        self.encoder_attn = MultiheadAttention(dim, heads, kdim=enc_dim, vdim=enc_dim,
                                               dropout=attn_dropout)
        self.encoder_attn_layer_norm = nn.LayerNorm(dim)
        self.fc1 = nn.Linear(dim, ffn)
        self.fc2 = nn.Linear(ffn, dim)
        self.final_layer_norm = nn.LayerNorm(dim)
        self.normalize_before = normalize_before
        self.p, self.pa = dropout, act_dropout
        self.site = ""

    def forward(self, x, enc, enc_pad_mask, self_attn_mask, self_pad_mask, need_attn=False):
        r = x
        if self.normalize_before:
            x = self.self_attn_layer_norm(x)
        x, _ = self.self_attn(x, x, x, key_padding_mask=self_pad_mask, attn_mask=self_attn_mask)
        x = r + drop(x, self.p, self.training, f"{self.site}/lin0", "tbc")
        if not self.normalize_before:
            x = self.self_attn_layer_norm(x)
        r = x
        if self.normalize_before:
            x = self.encoder_attn_layer_norm(x)
        x, attn = self.encoder_attn(x, enc, enc, key_padding_mask=enc_pad_mask,
                                    need_head_weights=need_attn)
        x = r + drop(x, self.p, self.training, f"{self.site}/lin1", "tbc")
        if not self.normalize_before:
            x = self.encoder_attn_layer_norm(x)
        r = x
        if self.normalize_before:
            x = self.final_layer_norm(x)
        x = drop(F.relu(self.fc1(x)), self.pa, self.training, f"{self.site}/lin2", "tbc")
        x = r + drop(self.fc2(x), self.p, self.training, f"{self.site}/lin3", "tbc")
        if not self.normalize_before:
            x = self.final_layer_norm(x)
        return x, attn


def future_mask(t: int) -> torch.Tensor:
    """-inf above the diagonal (s2st_transformer.py:465-477)."""
    return torch.triu(torch.full((t, t), float("-inf")), 1)


class Conv1dSubsampler(nn.Module):
    """s2st_transformer.py:94-140: (Conv1d k s2 p=k//2 -> GLU over channels) x n."""

    def __init__(self, in_ch, mid_ch, out_ch, kernel_sizes):
        super().__init__()
        n = len(kernel_sizes)
        self.n_layers = n
        self.conv_layers = nn.ModuleList(
            nn.Conv1d(in_ch if i == 0 else mid_ch // 2,
                      mid_ch if i < n - 1 else out_ch * 2, k, stride=2, padding=k // 2)
            for i, k in enumerate(kernel_sizes)
        )

    def out_lens(self, lens):
        out = lens.clone()
        for _ in range(self.n_layers):
            out = ((out.float() - 1) / 2 + 1).floor().long()
        return out

    def forward(self, x, lens):
        x = x.transpose(1, 2).contiguous()
        for c in self.conv_layers:
            x = F.glu(c(x), dim=1)
        return x.transpose(1, 2).transpose(0, 1).contiguous(), self.out_lens(lens)


class _PrenetLayers(nn.Module):
    """Tacotron2 Prenet: dropout applied ALWAYS, even in eval
    (fairseq/models/text_to_speech/tacotron2.py:85-98)."""

    def __init__(self, in_dim, n_layers, n_units, dropout):
        super().__init__()
        self.layers = nn.ModuleList(
            nn.Sequential(nn.Linear(in_dim if i == 0 else n_units, n_units), nn.ReLU())
            for i in range(n_layers)
        )
        self.dropout = dropout

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = drop(layer(x), self.dropout, True, f"dec.prenet/lin{i}", "btc")
        return x


class Postnet(nn.Module):
    """tacotron2.py:101-126: n x [Conv1d k p=(k-1)/2 -> BatchNorm1d -> tanh (not last) ->
    dropout].  BatchNorm uses batch statistics over ALL B x T positions in train mode."""

    def __init__(self, in_dim, n_channels, k, n_layers, dropout):
        super().__init__()
        self.convolutions = nn.ModuleList()
        for i in range(n_layers):
            cur = [
                nn.Conv1d(in_dim if i == 0 else n_channels,
                          n_channels if i < n_layers - 1 else in_dim, k, padding=(k - 1) // 2),
                nn.BatchNorm1d(n_channels if i < n_layers - 1 else in_dim),
            ] + ([nn.Tanh()] if i < n_layers - 1 else []) + [nn.Dropout(dropout)]
            self.convolutions.append(nn.Sequential(*cur))

    def forward(self, x):
        x = x.transpose(1, 2)
        for i, c in enumerate(self.convolutions):
            for m in c:  # (the Dropout module stays in the Sequential: state_dict indices as in the reference)
                x = drop(x, m.p, self.training, f"post/norm{i}", "bct") if isinstance(m, nn.Dropout) else m(x)
        return x.transpose(1, 2)


# ----------------------------------------------------------------------------------------
# model
# ----------------------------------------------------------------------------------------
class S2STEncoder(nn.Module):
    """S2STTransformerEncoder (s2st_transformer.py:143-256), fbank input (no HuBERT)."""

    def __init__(self, a):
        super().__init__()
        self.a = a
        self.middle_layers = [int(k) for k in a.middle_layers.split(",")]
        self.embed_scale = 1.0 if a.no_scale_embedding else math.sqrt(a.encoder_embed_dim)
        in_dim = a.hubert_hidden if getattr(a, "_hubert_input", False) else (
            a.input_feat_per_channel * a.input_channels)
        self.subsample = Conv1dSubsampler(
            in_dim, a.conv_channels, a.encoder_embed_dim,
            [int(k) for k in a.conv_kernel_sizes.split(",")])
        self.transformer_layers = nn.ModuleList(
            TransformerEncoderLayer(a.encoder_embed_dim, a.encoder_attention_heads,
                                    a.encoder_ffn_embed_dim, a.encoder_normalize_before,
                                    a.dropout, a.attention_dropout, a.activation_dropout)
            for _ in range(a.encoder_transformer_layers))
        self.layer_norm = nn.LayerNorm(a.encoder_embed_dim) if a.encoder_normalize_before else None
        self.aux_asr_norm = nn.LayerNorm(a.encoder_embed_dim) if a.asr_ce_weight > 0 else None
        self.aux_st_norm = nn.LayerNorm(a.encoder_embed_dim) if a.st_ce_weight > 0 else None
        self.embed_positions = _PositionalEmbeddingState()
        # speaker table: rows = len(args.speaker_to_id) of the flag's STRING (tasks/s2s_translation.py:156-160)
        n_spk = len(a.speaker_to_id) if getattr(a, "speaker_to_id", None) is not None else 0
        self.embed_speaker = nn.Embedding(n_spk, a.speaker_embed_dim) if n_spk else None

    def forward(self, src, src_lens, speaker=None):
        x, lens = self.subsample(src, src_lens)  # [E, B, C]
        x = self.embed_scale * x
        pad = lengths_to_padding_mask(lens, x.shape[0])
        x = x + positional_embedding(pad, x.shape[-1]).transpose(0, 1)
        if speaker is not None:  # s2st_transformer.py:203-206: every position, padded ones included
            x = x + self.embed_speaker(speaker).transpose(0, 1)
        x = drop(x, self.a.dropout, self.training, "enc.pe/rows0", "tbc")
        taps = []
        for i, layer in enumerate(self.transformer_layers):
            x = layer(x, pad)
            if i in self.middle_layers:
                taps.append(x)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        if self.aux_asr_norm is not None:
            taps[0] = self.aux_asr_norm(taps[0])
        if self.aux_st_norm is not None:
            taps[1] = self.aux_st_norm(taps[1])
        return {"encoder_out": x, "encoder_padding_mask": pad, "out_middle_layers": taps,
                "encoder_lens": lens}


class T2SEncoder(nn.Module):
    """T2STransformerEncoder (examples/s2s_trans/models/t2s_transformer.py:37-126): text input."""

    def __init__(self, a):
        super().__init__()
        self.a = a
        C = a.encoder_embed_dim
        self.embed_tokens = nn.Embedding(a.src_vocab_size, C, padding_idx=PAD)
        k = a.encoder_conv_kernel_size
        self.prenet = nn.ModuleList(
            nn.Sequential(nn.Conv1d(C, C, kernel_size=k, padding=(k - 1) // 2), nn.BatchNorm1d(C), nn.ReLU(),
                          nn.Dropout(a.encoder_dropout))
            for _ in range(a.encoder_conv_layers))
        self.prenet_proj = nn.Linear(C, C)
        self.pos_emb_alpha = nn.Parameter(torch.ones(1))
        self.transformer_layers = nn.ModuleList(
            TransformerEncoderLayer(C, a.encoder_attention_heads, a.encoder_ffn_embed_dim, a.encoder_normalize_before,
                                    a.dropout, a.attention_dropout, a.activation_dropout)
            for _ in range(a.encoder_transformer_layers))
        self.layer_norm = nn.LayerNorm(C) if a.encoder_normalize_before else None
        self.embed_positions = _PositionalEmbeddingState()
        # t2s_transformer.py:40-46 (+ tasks/s2s_translation_mtl.py:133-150: Embedding(len(speaker_to_id), speaker_embed_dim),
        # the length of the flag's STRING)
        n_spk = len(a.speaker_to_id) if getattr(a, "speaker_to_id", None) is not None else 0
        self.embed_speaker = nn.Embedding(n_spk, a.speaker_embed_dim) if n_spk else None
        self.spk_emb_proj = nn.Linear(C + a.speaker_embed_dim, C) if n_spk else None

    def forward(self, src_tokens, src_lens, speaker=None):
        x = self.embed_tokens(src_tokens).transpose(1, 2).contiguous()
        for i, conv in enumerate(self.prenet):
            for m in conv:
                x = drop(x, m.p, self.training, f"enc.prenet/norm{i}", "bct") if isinstance(m, nn.Dropout) else m(x)
        x = self.prenet_proj(x.transpose(1, 2).contiguous())
        pad = src_tokens.eq(PAD)
        x = x + self.pos_emb_alpha * positional_embedding(pad, x.shape[-1])
        x = drop(x, self.a.dropout, self.training, "enc.pe/rows0", "btc").transpose(0, 1)
        for layer in self.transformer_layers:
            x = layer(x, pad)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        if self.embed_speaker is not None:  # t2s_transformer.py:107-111: every position, padded ones included
            emb = self.embed_speaker(speaker).transpose(0, 1).expand(x.shape[0], x.shape[1], -1)
            x = self.spk_emb_proj(torch.cat([x, emb], dim=2))
        return {"encoder_out": x, "encoder_padding_mask": pad, "out_middle_layers": [],
                "encoder_lens": (~pad).long().sum(1)}


class S2STDecoder(nn.Module):
    """S2STTransformerDecoder (s2st_transformer.py:319-477), teacher-forced path."""

    def __init__(self, a):
        super().__init__()
        self.a = a
        self.out_dim = a.output_frame_dim * a.n_frames_per_step
        self.pos_emb_alpha = nn.Parameter(torch.ones(1))
        self.prenet = nn.Sequential(
            _PrenetLayers(self.out_dim, a.prenet_layers, a.prenet_dim, a.prenet_dropout),
            nn.Linear(a.prenet_dim, a.decoder_embed_dim))
        self.transformer_layers = nn.ModuleList(
            TransformerDecoderLayer(a.decoder_embed_dim, a.decoder_attention_heads,
                                    a.decoder_ffn_embed_dim, a.encoder_embed_dim,
                                    a.decoder_normalize_before, a.dropout,
                                    a.attention_dropout, a.activation_dropout)
            for _ in range(a.decoder_transformer_layers))
        self.layer_norm = nn.LayerNorm(a.decoder_embed_dim) if a.decoder_normalize_before else None
        self.feat_proj = nn.Linear(a.decoder_embed_dim, self.out_dim)
        self.eos_proj = nn.Linear(a.decoder_embed_dim, 1)
        self.postnet = Postnet(self.out_dim, a.postnet_conv_dim, a.postnet_conv_kernel_size,
                               a.postnet_layers, a.postnet_dropout)
        # (t2s_transformer.py:168-170: the text-input model's head reads feature_out, width out_dim)
        self.ctc_proj = (nn.Linear(self.out_dim if getattr(a, "text_encoder", False) else a.encoder_embed_dim,
                                   a.src_vocab_size) if a.ctc_weight > 0 else None)
        # s2st_transformer_mtl.py:266-271: a second CTC head (target text) on a decoder layer's output
        self.ctc_proj_tgt = (nn.Linear(a.decoder_embed_dim, a.tgt_vocab_size)
                             if getattr(a, "ctc_weight_tgt", 0.0) > 0 else None)
        self.middle_layers_decoder = [int(k) for k in str(getattr(a, "middle_layers_decoder", "6")).split(",")]
        self.embed_positions = _PositionalEmbeddingState()
        n_spk = len(a.speaker_to_id) if getattr(a, "speaker_to_id", None) is not None else 0
        if getattr(a, "text_encoder", False):
            n_spk = 0  # T2STransformerDecoder has no speaker table; its `speaker` argument is unused (t2s_transformer.py:172-176)
        self.embed_speaker = nn.Embedding(n_spk, a.speaker_embed_dim_dec) if n_spk else None

    def forward(self, prev, enc, target_lengths, speaker=None):
        if speaker is not None and self.embed_speaker is not None:  # s2st_transformer.py:441-444: the speaker row replaces the first input frame
            prev = torch.cat([self.embed_speaker(speaker), prev[:, 1:, :]], 1)
        pad = lengths_to_padding_mask(target_lengths, prev.shape[1])
        pos = positional_embedding(pad, self.a.decoder_embed_dim)
        x = self.prenet(prev)
        x = x + self.pos_emb_alpha * pos
        x = drop(x, self.a.dropout, self.training, "dec.pe/rows0", "btc")
        x = x.transpose(0, 1)
        self_pad = pad if bool(pad.any()) else None
        enc_pad = enc["encoder_padding_mask"] if bool(enc["encoder_padding_mask"].any()) else None
        fm = future_mask(x.shape[0])
        attn = None
        taps_dec = []
        n = len(self.transformer_layers)
        for i, layer in enumerate(self.transformer_layers):
            x, a_ = layer(x, enc["encoder_out"], enc_pad, fm, self_pad, need_attn=(i == n - 1))
            if i in self.middle_layers_decoder:
                taps_dec.append(x)  # raw layer output, [D, B, C] (s2st_transformer_mtl.py:325-327)
            if a_ is not None:
                attn = a_
        if attn is not None:
            attn = attn.mean(dim=0).transpose(2, 1)  # [B, E, D]
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        x = x.transpose(0, 1)
        feat = self.feat_proj(x)
        eos = self.eos_proj(x)
        post = feat + self.postnet(feat)
        return post, eos, {"attn": attn, "feature_out": feat,
                           "out_middle_layers": enc["out_middle_layers"], "out_middle_layers_decoder": taps_dec}


class AuxTextDecoder(nn.Module):
    """ASR/STTransformerDecoderScriptable (s2st_transformer.py:483-578) over
    TransformerDecoderBase (fairseq/models/transformer/transformer_decoder.py:253-378).

    Dims per SURVEY.md Appendix A.2: embed_tokens (V, in_dim), project_in (d, in_dim) if
    d != in_dim, layers at d with FFN = main decoder FFN and cross K/V from 512,
    final LN if normalize_before, project_out (512, d) no bias when d != 512,
    output_projection (V, 512) no bias."""

    def __init__(self, a, vocab, in_dim, d, n_layers, tap, out_dim=512):
        """``tap`` None / ``out_dim`` = d: the s2t model's own decoder (oracle/s2t_oracle.py) -- it reads the encoder's final
        output and projects at its own width (s2t_transformer_me.py:527-529)."""
        super().__init__()
        self.a, self.d, self.tap = a, d, tap
        self.embed_tokens = nn.Embedding(vocab, in_dim, padding_idx=PAD)
        self.embed_scale = 1.0 if a.no_scale_embedding else math.sqrt(d)
        self.project_in_dim = nn.Linear(in_dim, d, bias=False) if d != in_dim else None
        self.layers = nn.ModuleList(
            TransformerDecoderLayer(d, a.decoder_attention_heads, a.decoder_ffn_embed_dim,
                                    a.encoder_embed_dim, a.decoder_normalize_before,
                                    a.dropout, a.attention_dropout, a.activation_dropout)
            for _ in range(n_layers))
        self.layer_norm = nn.LayerNorm(d) if a.decoder_normalize_before else None
        # (aux heads: DecoderConfig.output_dim default 512, transformer_config.py:63-68)
        self.out_dim = out_dim
        self.project_out_dim = nn.Linear(d, out_dim, bias=False) if d != out_dim else None
        self.output_projection = nn.Linear(out_dim, vocab, bias=False)
        self.register_buffer("version", torch.Tensor([3]))
        self.embed_positions = _PositionalEmbeddingState()
        self.site = ""

    def forward(self, tokens, enc):
        x = self.embed_scale * self.embed_tokens(tokens)
        if self.project_in_dim is not None:
            x = self.project_in_dim(x)
        x = x + positional_embedding(tokens, self.d)
        x = drop(x, self.a.dropout, self.training, f"{self.site}.pe/rows0", "btc")
        x = x.transpose(0, 1)
        pad = tokens.eq(PAD)
        self_pad = pad if bool(pad.any()) else None
        enc_pad = enc["encoder_padding_mask"] if bool(enc["encoder_padding_mask"].any()) else None
        fm = future_mask(x.shape[0])
        for layer in self.layers:
            x, _ = layer(x, enc["encoder_out"] if self.tap is None else enc["out_middle_layers"][self.tap], enc_pad, fm, self_pad)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        x = x.transpose(0, 1)
        if self.project_out_dim is not None:
            x = self.project_out_dim(x)
        return self.output_projection(x)


class S2STModel(nn.Module):
    """S2STTransformerModel (s2st_transformer.py:580-789)."""

    def __init__(self, a):
        super().__init__()
        self.a = a
        self.encoder = T2SEncoder(a) if getattr(a, "text_encoder", False) else S2STEncoder(a)
        self.decoder = S2STDecoder(a)
        # reference mutates args.decoder_embed_dim while building the aux decoders
        # (:492-493, :541-542): the ASR embedding is built at the main decoder dim, the ST
        # embedding at the ASR decoder dim.
        self.aux_asr_decoder = None
        self.aux_st_decoder = None
        cur = a.decoder_embed_dim
        if a.asr_ce_weight > 0:
            self.aux_asr_decoder = AuxTextDecoder(a, a.src_vocab_size, cur,
                                                  a.asr_decoder_embed_dim,
                                                  a.asr_decoder_layers, tap=0)
            cur = a.asr_decoder_embed_dim
        if a.st_ce_weight > 0:
            self.aux_st_decoder = AuxTextDecoder(a, a.tgt_vocab_size, cur,
                                                 a.st_decoder_embed_dim,
                                                 a.st_decoder_layers, tap=1)

    def forward(self, src_tokens, src_lengths, prev_output_tokens, target_lengths,
                prev_src_text_tokens=None, prev_tgt_text_tokens=None, speaker=None):
        enc = self.encoder(src_tokens, src_lengths, speaker=speaker) if speaker is not None else self.encoder(src_tokens, src_lengths)
        dec = (self.decoder(prev_output_tokens, enc, target_lengths, speaker=speaker) if speaker is not None
               else self.decoder(prev_output_tokens, enc, target_lengths))
        asr = st = None
        if self.aux_asr_decoder is not None:
            asr = self.aux_asr_decoder(prev_src_text_tokens, enc)
        if self.aux_st_decoder is not None:
            st = self.aux_st_decoder(prev_tgt_text_tokens, enc)
        return dec, asr, st, enc


# ----------------------------------------------------------------------------------------
# criterion (examples/s2s_trans/criterions/s2st_loss.py)
# ----------------------------------------------------------------------------------------
USE_TORCH_CTC = False  # bench.py's CPU-baseline leg flips this: same numbers, ATen's C++ loop


def ctc_loss_mean(lprobs, targets_flat, input_lens, target_lens, blank=0):
    """CTC negative log-likelihood, alpha recursion in log space, reduction='mean'
    (per-utterance loss / clamp(target_len, 1), then batch mean) and zero_infinity=True --
    the semantics of ``torch.nn.CTCLoss(reduction="mean", zero_infinity=True)`` as used at
    s2st_loss.py:173-176, 242-243.  Graves et al. 2006.  lprobs [T, B, V]."""
    if USE_TORCH_CTC:
        return F.ctc_loss(lprobs, targets_flat, input_lens, target_lens, blank=blank,
                          reduction="mean", zero_infinity=True)
    T, B, V = lprobs.shape
    losses = []
    off = 0
    NEG = -1.0e30  # finite stand-in for log(0): keeps autograd free of inf*0 NaNs
    for b in range(B):
        L = int(target_lens[b])
        Tb = int(input_lens[b])
        tgt = targets_flat[off:off + L]
        off += L
        S = 2 * L + 1
        ext = torch.full((S,), blank, dtype=torch.long)
        ext[1::2] = tgt
        lp = lprobs[:Tb, b, :]
        init = torch.full((S,), NEG, dtype=lprobs.dtype)
        a0 = [lp[0, blank]] + ([lp[0, ext[1]]] if S > 1 else [])
        alpha = torch.cat([torch.stack(a0), init[len(a0):]])
        can_skip = torch.zeros(S, dtype=torch.bool)
        if S > 2:
            can_skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        for t in range(1, Tb):
            a1 = torch.cat([alpha.new_full((1,), NEG), alpha[:-1]])
            a2 = torch.cat([alpha.new_full((2,), NEG), alpha[:-2]])
            a2 = torch.where(can_skip, a2, torch.full_like(a2, NEG))
            m = torch.maximum(torch.maximum(alpha, a1), a2).detach()
            s = torch.exp(alpha - m) + torch.exp(a1 - m) + torch.exp(a2 - m)
            alpha = torch.clamp(torch.log(s) + m + lp[t, ext], min=NEG)
        if S > 1:
            m = torch.maximum(alpha[-1], alpha[-2]).detach()
            tot = torch.log(torch.exp(alpha[-1] - m) + torch.exp(alpha[-2] - m)) + m
        else:
            tot = alpha[-1]
        loss = -tot
        if float(tot.detach()) < -1.0e29:  # infeasible alignment -> zero_infinity
            loss = loss * 0.0
        losses.append(loss / max(L, 1))
    return torch.stack(losses).mean()


def label_smoothed_nll_loss(lprobs, target, eps, ignore_index=PAD):
    """s2st_loss.py:33-50 (reduce=True)."""
    target = target.unsqueeze(-1)
    nll = -lprobs.gather(dim=-1, index=target)
    smooth = -lprobs.sum(dim=-1, keepdim=True)
    pm = target.eq(ignore_index)
    nll = nll.masked_fill(pm, 0.0).sum()
    smooth = smooth.masked_fill(pm, 0.0).sum()
    eps_i = eps / (lprobs.size(-1) - 1)
    return (1.0 - eps - eps_i) * nll + eps_i * smooth, nll


def guided_attention_loss(attn, src_lens, tgt_lens, sigma):
    """s2st_loss.py:106-144 (mean over valid (t, s) cells); attn [B, E, D]."""
    B = attn.shape[0]
    ms, mt = int(src_lens.max()), int(tgt_lens.max())
    w = torch.zeros(B, mt, ms)
    for i in range(B):
        s, t = int(src_lens[i]), int(tgt_lens[i])
        gx, gy = torch.meshgrid(torch.arange(t), torch.arange(s), indexing="ij")
        w[i, :t, :s] = 1.0 - torch.exp(-((gy.float() / s - gx.float() / t) ** 2) / (2 * sigma ** 2))
    mask = (~lengths_to_padding_mask(tgt_lens, mt)).unsqueeze(2) & \
           (~lengths_to_padding_mask(src_lens, ms)).unsqueeze(1)
    return (w * attn.transpose(1, 2)).masked_select(mask).mean()


def ctc_input_lengths(src_lens, kernel_sizes):
    """s2st_loss.py:231-232."""
    out = src_lens.clone()
    for k in kernel_sizes:
        out = (out - k + 2 * (k // 2)) // 2 + 1
    return out


def criterion_forward(model: S2STModel, sample: Dict, a=None):
    """Tacotron2Criterion.forward (s2st_loss.py:179-292), reduction='mean'.
    Returns (loss, sample_size, logging_output(dict of tensors/ints), net outputs)."""
    a = model.a if a is None else a
    tgt = sample["tgt_speech"]
    B, D, _ = tgt.shape
    tl = sample["target_lengths"]
    eos_tgt = (torch.arange(D).view(1, D).expand(B, -1) == (tl.view(B, 1) - 1)).float()
    ni = sample["net_input"]
    text_in = getattr(a, "text_encoder", False)  # t2s_loss.py:110-121: src_tokens = src_text
    (post, eos, extra), asr, st, enc = model(
        sample["src_text"] if text_in else ni["src_speech"],
        sample["src_text_len"] if text_in else ni["src_speech_lens"], ni["prev_output_tokens"], tl,
        ni.get("prev_src_text_tokens") if a.asr_ce_weight > 0 else None,
        ni.get("prev_tgt_text_tokens") if a.st_ce_weight > 0 else None,
        **({"speaker": sample["speaker"]} if sample.get("speaker") is not None else {}))  # s2st_loss.py:217
    mask = ~lengths_to_padding_mask(tl, D)
    _eos = eos[mask].squeeze(-1)
    _et = eos_tgt[mask]
    _ft = tgt[mask]
    _fo = extra["feature_out"][mask]
    _fp = post[mask]
    l1 = F.l1_loss(_fo, _ft) + F.l1_loss(_fp, _ft)
    mse = F.mse_loss(_fo, _ft) + F.mse_loss(_fp, _ft)
    eos_loss = F.binary_cross_entropy_with_logits(
        _eos, _et, pos_weight=torch.tensor(a.bce_pos_weight))
    zero = torch.zeros(())
    attn_loss = zero
    if a.use_guided_attention_loss:
        attn_loss = guided_attention_loss(extra["attn"], enc["encoder_lens"], tl,
                                          a.guided_attention_loss_sigma)
    ctc = zero
    lprobs_ctc = None
    if a.ctc_weight > 0 and text_in:  # t2s_loss.py:134-144: source text against the decoder's feature_out
        lprobs_ctc = F.log_softmax(model.decoder.ctc_proj(extra["feature_out"]).float(), dim=-1).transpose(0, 1)  # [D, B, V]
        smask = ~lengths_to_padding_mask(sample["src_text_len"], sample["src_text"].shape[1])
        ctc = ctc_loss_mean(lprobs_ctc, sample["src_text"].masked_select(smask), tl, sample["src_text_len"]) * a.ctc_weight
    elif a.ctc_weight > 0:
        ilens = ctc_input_lengths(ni["src_speech_lens"],
                                  [int(k) for k in a.conv_kernel_sizes.split(",")])
        logits = model.decoder.ctc_proj(extra["out_middle_layers"][0].transpose(0, 1))
        lprobs_ctc = F.log_softmax(logits.float(), dim=-1).transpose(0, 1)  # [E, B, V]
        smask = ~lengths_to_padding_mask(sample["src_text_len"], sample["src_text"].shape[1])
        flat = sample["src_text"].masked_select(smask)
        ctc = ctc_loss_mean(lprobs_ctc, flat, ilens, sample["src_text_len"]) * a.ctc_weight
    ctc_tgt = zero
    if getattr(a, "ctc_weight_tgt", 0.0) > 0:  # s2st_loss_mtl.py:171-186
        logits = model.decoder.ctc_proj_tgt(extra["out_middle_layers_decoder"][0].transpose(0, 1))
        lp_t = F.log_softmax(logits.float(), dim=-1).transpose(0, 1)  # [D, B, V]
        tmask = ~lengths_to_padding_mask(sample["tgt_text_len"], sample["tgt_text"].shape[1])
        ctc_tgt = ctc_loss_mean(lp_t, sample["tgt_text"].masked_select(tmask), tl, sample["tgt_text_len"]) * a.ctc_weight_tgt
    log = {}
    asr_loss = st_loss = zero
    if a.asr_ce_weight > 0:
        lp = F.log_softmax(asr.float(), dim=-1)
        l, _ = label_smoothed_nll_loss(lp.view(-1, lp.size(-1)), sample["src_text"].view(-1),
                                       a.label_smoothing)
        asr_loss = l / sample["src_txt_ntokens"] * a.asr_ce_weight
        m = sample["src_text"].view(-1).ne(PAD)
        log["asr_n_correct"] = int((lp.view(-1, lp.size(-1)).argmax(1)[m]
                                    == sample["src_text"].view(-1)[m]).sum())
        log["asr_total"] = int(m.sum())
    if a.st_ce_weight > 0:
        lp = F.log_softmax(st.float(), dim=-1)
        l, _ = label_smoothed_nll_loss(lp.view(-1, lp.size(-1)), sample["tgt_text"].view(-1),
                                       a.label_smoothing)
        st_loss = l / sample["tgt_txt_ntokens"] * a.st_ce_weight
        m = sample["tgt_text"].view(-1).ne(PAD)
        log["st_n_correct"] = int((lp.view(-1, lp.size(-1)).argmax(1)[m]
                                   == sample["tgt_text"].view(-1)[m]).sum())
        log["st_total"] = int(m.sum())
    l1, mse, eos_loss, attn_loss = (l1 * a.l1_loss_weight, mse * a.mse_loss_weight,
                                    eos_loss * a.eos_loss_weight, attn_loss * a.attn_loss_weight)
    loss = l1 + mse + eos_loss + attn_loss + ctc + ctc_tgt + asr_loss + st_loss
    log.update({
        "ctc_loss_tgt": ctc_tgt.detach(),
        "loss": loss.detach(), "ntokens": sample["ntokens"], "nsentences": sample["nsentences"],
        "sample_size": sample["ntokens"], "l1_loss": l1.detach(), "mse_loss": mse.detach(),
        "eos_loss": eos_loss.detach(), "attn_loss": attn_loss.detach(),
        "ctc_loss": ctc.detach(), "aux_asr_loss": asr_loss.detach(),
        "aux_st_loss": st_loss.detach(),
    })
    outs = {"post_feat_out": post, "eos_out": eos, "feature_out": extra["feature_out"],
            "attn": extra["attn"], "encoder_out": enc["encoder_out"],
            "taps": enc["out_middle_layers"], "asr_logits": asr, "st_logits": st,
            "ctc_lprobs": lprobs_ctc, "encoder_lens": enc["encoder_lens"]}
    return loss, sample["ntokens"], log, outs


# ----------------------------------------------------------------------------------------
# integer outputs defined by the build
# ----------------------------------------------------------------------------------------
def ctc_greedy_path(lprobs, input_lens):
    """Frame-wise arg-max of the CTC log-probs [E, B, V] -> int64 [B, E], -1 past the
    utterance's input length."""
    am = lprobs.argmax(-1).transpose(0, 1).clone()
    am[lengths_to_padding_mask(input_lens, am.shape[1])] = -1
    return am


def stop_indices(eos_logits, threshold=0.5):
    """First decoder step with sigmoid(eos) > threshold, or D if none
    (stop rule of fairseq/speech_generator_for_s2st.py:89-96 applied to teacher-forced
    logits [B, D, 1])."""
    hit = torch.sigmoid(eos_logits.squeeze(-1)) > threshold
    D = hit.shape[1]
    idx = torch.where(hit, torch.arange(D).expand_as(hit), torch.full_like(hit, D, dtype=torch.long))
    return idx.min(dim=1).values


# ----------------------------------------------------------------------------------------
# train-step arithmetic (fairseq/trainer.py:838-873)
# ----------------------------------------------------------------------------------------
def inverse_sqrt_lr(num_updates, lr, warmup_updates, warmup_init_lr=-1.0):
    """fairseq/optim/lr_scheduler/inverse_square_root_schedule.py:52-85."""
    if warmup_init_lr < 0:
        warmup_init_lr = 0.0 if warmup_updates > 0 else lr
    if num_updates < warmup_updates:
        return warmup_init_lr + num_updates * (lr - warmup_init_lr) / warmup_updates
    return lr * warmup_updates ** 0.5 * num_updates ** -0.5


@torch.no_grad()
def clip_grad_norm_(params, max_norm):
    """fairseq/utils.py:345-395: norm of per-tensor fp32 norms; g *= min(1, c/(n+1e-6))."""
    grads = [p.grad for p in params if p.grad is not None]
    total = torch.norm(torch.stack([torch.norm(g, p=2, dtype=torch.float32) for g in grads]))
    if max_norm > 0:
        coef = (float(max_norm) / (total + 1e-6)).clamp_(max=1)
        for g in grads:
            g.mul_(coef)
    return total


class FairseqAdam:
    """fairseq/optim/adam.py:163-239: denom = sqrt(v) + eps (before bias correction),
    step = lr * sqrt(1-b2^t) / (1-b1^t); decoupled weight decay."""

    def __init__(self, params, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = list(params)
        self.b1, self.b2 = betas
        self.eps, self.wd = eps, weight_decay
        self.state = {}

    @torch.no_grad()
    def step(self, lr):
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.setdefault(p, {"step": 0, "m": torch.zeros_like(p),
                                           "v": torch.zeros_like(p)})
            st["step"] += 1
            g = p.grad
            st["m"].mul_(self.b1).add_(g, alpha=1 - self.b1)
            st["v"].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = st["v"].sqrt().add_(self.eps)
            bc1 = 1 - self.b1 ** st["step"]
            bc2 = 1 - self.b2 ** st["step"]
            if self.wd != 0:
                p.add_(p, alpha=-self.wd * lr)
            p.addcdiv_(st["m"], denom, value=-(lr * math.sqrt(bc2) / bc1))


def train_step(model, opt: FairseqAdam, sample, num_updates, lr, warmup_updates, clip_norm,
               world_size=1):
    """One optimizer update as Trainer.train_step does it (fairseq/trainer.py:709-1010):
    fwd, bwd, grads *= world/sample_size, clip, Adam with lr = schedule(num_updates)."""
    model.train()
    for p in model.parameters():
        p.grad = None
    loss, sample_size, log, outs = criterion_forward(model, sample)
    loss.backward()
    with torch.no_grad():
        for p in model.parameters():
            if p.grad is not None:
                p.grad.mul_(world_size / float(sample_size))
    gnorm = clip_grad_norm_(list(model.parameters()), clip_norm)
    cur_lr = inverse_sqrt_lr(num_updates, lr, warmup_updates)
    opt.step(cur_lr)
    return loss.detach(), gnorm, cur_lr, log, outs
