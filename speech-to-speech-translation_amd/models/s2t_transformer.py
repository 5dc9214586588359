"""``--arch s2t_transformer_hubert`` on the MI355X engine: the ST / ASR pre-training stage of the mix- / prompt-tuning
recipes (examples/s2s_trans/run_mix_tuning.sh:92-118: ``--task s2s_translation --criterion s2t_loss --arch
s2t_transformer_hubert``), whose checkpoint the s2st stage then reads with ``--load-pretrained-encoder-from``
(run_mix_tuning.sh:143).

Host-side mirror of fairseq/models/speech_to_text/s2t_transformer_me.py:82-330: the speech encoder of ``s2st_transformer``
(conv subsampler over fbank or frozen-HuBERT features, sinusoidal positions, pre-LN Transformer layers, final LayerNorm --
the same parameter names, :333-420) and, in place of the mel decoder, ONE fairseq ``TransformerDecoder`` over the TARGET
dictionary (:266-283).  The engine's ``s2t_mode`` switch builds that pair out of the shared kernels (include/s2st_hip.h);
the decoder's parameter names are fairseq's (``decoder.embed_tokens / layers.N / layer_norm / output_projection``).
"""
from __future__ import annotations

import torch

from ..registry import register_model, register_model_architecture
from .s2st_transformer import S2STTransformerModel, base_architecture, _register


@register_model("s2t_transformer_hubert")
class S2TTransformerModel(S2STTransformerModel):
    @staticmethod
    def add_args(parser):
        """s2t_transformer_me.py:96-243 (the flags the engine honours; ``--ctc-loss``, ``--share-decoder-input-output-embed``,
        ``--layernorm-embedding`` and ``--encoder-freezing-updates`` > 0 are refused in build_model)."""
        a = parser.add_argument
        a("--conv-kernel-sizes", type=str)
        a("--conv-channels", type=int)
        a("--hubert-hidden", type=int, default=768)
        a("--activation-fn", type=str, default="relu")
        for f in ("--dropout", "--attention-dropout", "--activation-dropout"):
            a(f, type=float)
        for f in ("--encoder-embed-dim", "--encoder-ffn-embed-dim", "--encoder-layers", "--encoder-attention-heads",
                  "--decoder-embed-dim", "--decoder-ffn-embed-dim", "--decoder-layers", "--decoder-attention-heads",
                  "--encoder-freezing-updates"):
            a(f, type=int)
        for f in ("--encoder-normalize-before", "--decoder-normalize-before", "--share-decoder-input-output-embed",
                  "--layernorm-embedding", "--no-scale-embedding", "--ctc-loss"):
            a(f, action="store_true")
        a("--load-pretrained-encoder-from", type=str)
        a("--load-pretrained-hubert-from", type=str)

    @classmethod
    def build_model(cls, args, task):
        s2t_architecture(args)
        for flag in ("share_decoder_input_output_embed", "layernorm_embedding", "ctc_loss"):
            if getattr(args, flag, False):
                raise NotImplementedError(f"--{flag.replace('_', '-')} is not built on the HIP path (the recipes do not use it)")
        if int(getattr(args, "encoder_freezing_updates", 0) or 0) > 0:
            raise NotImplementedError("--encoder-freezing-updates > 0 is not built on the HIP path (the recipes use 0)")
        args.s2t_mode = True
        # the decoder -- embedding AND output projection -- is built over task.target_dictionary whatever --test-type says
        # (s2t_transformer_me.py:268-283)
        args.src_vocab_size = len(task.source_dictionary)
        args.tgt_vocab_size = len(task.target_dictionary)
        return super().build_model(args, task)

    def __init__(self, args, device, precise: bool = False):
        super().__init__(args, device, precise)
        # (fairseq's TransformerDecoder carries a version buffer like the aux decoders': transformer_decoder.py:91)
        _register(self, "decoder.version", torch.tensor([3.0], device=device), True)
        # --test-type is a flag of the CRITERION (s2t_loss.py:31-34) in the namespace model and criterion share: batches are
        # prepared for the device (Engine.prepare picks the decoder's text) before the criterion sees them
        self.test_type = self.engine.s2t_test_type = str(getattr(args, "test_type", "asr") or "asr")

    def _text_sample(self, src_tokens, src_lengths, prev_output_tokens, target=None):
        B, L = prev_output_tokens.shape
        key = "src" if self.test_type == "asr" else "tgt"
        dummy = torch.zeros(B, 1, max(int(self.engine.cfg.out_dim), 4))
        s = {"net_input": {"src_speech": src_tokens, "src_speech_lens": src_lengths, "prev_output_tokens": dummy,
                           f"prev_{key}_text_tokens": prev_output_tokens},
             "target_lengths": torch.ones(B, dtype=torch.long), "ntokens": B,
             f"{key}_text_len": prev_output_tokens.ne(1).sum(1)}
        if target is not None:
            s[f"{key}_text"] = target
        return s

    def forward(self, src_tokens, src_lengths, collated_audios, padding_mask, prev_output_tokens):
        """s2t_transformer_me.py:308-330: ``decoder(prev_output_tokens, encoder(src...))`` -> ``(logits [B, L, V], None)``."""
        src_tokens, src_lengths = self._front_end(src_tokens, src_lengths, collated_audios, padding_mask)
        self.engine.s2t_test_type = self.test_type
        o = self.engine.forward(self._text_sample(src_tokens, src_lengths, prev_output_tokens), training=self.training,
                                want_attn=False, with_loss=False)
        return o["asr_logits"], None

    def forward_encoder(self, src_tokens, src_lengths, collated_audios=None, padding_mask=None, **kwargs):
        src_tokens, src_lengths = self._front_end(src_tokens, src_lengths, collated_audios, padding_mask)
        B = src_tokens.shape[0]
        s = {"net_input": {"src_speech": src_tokens, "src_speech_lens": src_lengths,
                           "prev_output_tokens": torch.zeros(B, 1, max(int(self.engine.cfg.out_dim), 4))},
             "target_lengths": torch.ones(B, dtype=torch.long), "ntokens": B}
        o = self.engine.forward(s, training=self.training, want_attn=False, with_loss=False)
        lens = o["encoder_lens"].long()
        E = o["encoder_out"].shape[1]
        pad = torch.arange(E, device=lens.device).unsqueeze(0) >= lens.unsqueeze(1)
        return {"encoder_out": [o["encoder_out"].transpose(0, 1)], "encoder_padding_mask": [pad] if bool(pad.any()) else [],
                "encoder_embedding": [], "encoder_states": [], "src_tokens": [], "src_lengths": []}

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        """s2t_transformer_me.py:285-295: (log-)softmax of the decoder's logits, batch first."""
        logits = net_output[0]
        B, L, V = logits.shape
        from ..runtime import binding as bd
        out = torch.empty(B * L, V, dtype=torch.float32, device=logits.device)
        bd.call("s2st_log_softmax_rows_f32", logits.contiguous().view(B * L, V), V, out, V, B * L, V, 1 if log_probs else 0)
        out = out.view(B, L, V)
        out.batch_first = True
        return out

    def get_targets(self, sample, test_type, net_output):
        return sample["src_text"] if test_type == "asr" else sample["tgt_text"]


@register_model_architecture("s2t_transformer_hubert", "s2t_transformer_hubert")
def s2t_architecture(args):
    """Defaults of s2t_transformer_me.py:493-533 (8 heads, pre-LN on both sides, activation / attention dropout = dropout,
    and the same ``conv_chaFnnels`` typo that pins the subsampler to 1024 channels), mapped onto the flag names the shared
    engine configuration reads."""
    def g(k, v):
        if getattr(args, k, None) is None:
            setattr(args, k, v)

    g("encoder_freezing_updates", 0)
    g("conv_kernel_sizes", "5,5")
    args.conv_channels = 1024
    g("encoder_embed_dim", 512)
    g("encoder_ffn_embed_dim", 2048)
    g("encoder_layers", 12)
    g("encoder_attention_heads", 8)
    if getattr(args, "encoder_normalize_before", None) is None:
        args.encoder_normalize_before = True
    g("decoder_embed_dim", args.encoder_embed_dim)
    g("decoder_ffn_embed_dim", args.encoder_ffn_embed_dim)
    g("decoder_layers", 6)
    g("decoder_attention_heads", 8)
    if getattr(args, "decoder_normalize_before", None) is None:
        args.decoder_normalize_before = True
    g("dropout", 0.1)
    g("attention_dropout", args.dropout)
    g("activation_dropout", args.dropout)
    g("activation_fn", "relu")
    g("no_scale_embedding", False)
    # the shared engine configuration's names for the two depths
    args.encoder_transformer_layers = args.encoder_layers
    args.decoder_transformer_layers = args.decoder_layers
    # nothing of the mel decoder exists in this model: its flags only have to be well-formed
    args.asr_ce_weight = args.st_ce_weight = args.ctc_weight = 0.0
    args.prenet_layers, args.postnet_layers = 0, 0
    return base_architecture(args)


@register_model_architecture("s2t_transformer_hubert", "s2t_transformer_hubert_s")
def s2t_architecture_s(args):  # s2t_transformer_me.py:536-543
    def g(k, v):
        if getattr(args, k, None) is None:
            setattr(args, k, v)
    g("encoder_embed_dim", 256)
    g("encoder_ffn_embed_dim", 256 * 8)
    g("encoder_attention_heads", 4)
    g("decoder_attention_heads", 4)
    g("dropout", 0.1)
    return s2t_architecture(args)
