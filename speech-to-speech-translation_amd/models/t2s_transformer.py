"""``--arch t2s_transformer`` on the MI355X engine.

Host-side mirror of examples/s2s_trans/models/t2s_transformer.py:279-371 (Transformer-TTS, arXiv 1809.08895): a TEXT
encoder -- token embedding, ``--encoder-conv-layers`` x (Conv1d k5 + BatchNorm1d + ReLU + dropout), a linear
projection, ``pos_emb_alpha``-scaled sinusoidal positions, Transformer layers (post-LN by default) -- in front of the
mel decoder of ``s2st_transformer``.  The engine's ``text_input`` switch builds that front (include/s2st_hip.h);
decoder, losses and optimizer are the shared kernels.  The optional decoder-side CTC head (``ctc_proj`` over
``feature_out``, :161-163, 258-263) and the speaker conditioning of the encoder (``spk_emb_proj(cat[x, embed_speaker(speaker)])``
after the last layer, :43-46, 107-111; table ``Embedding(len(--speaker-to-id), --speaker-embed-dim)`` as the reference's
``task.get_speaker_embeddings(args)`` builds it) are engine switches too.
"""
from __future__ import annotations

from ..registry import register_model, register_model_architecture
from .s2st_transformer import S2STTransformerModel, base_architecture


@register_model("t2s_transformer")
class T2STransformerModel(S2STTransformerModel):
    @staticmethod
    def add_args(parser):
        S2STTransformerModel.add_args(parser)
        a = parser.add_argument
        a("--encoder-dropout", type=float)
        a("--encoder-conv-layers", type=int)
        a("--encoder-conv-kernel-size", type=int)

    @classmethod
    def build_model(cls, args, task):
        t2s_architecture(args)
        if getattr(args, "asr_ce_weight", 0.0) or getattr(args, "st_ce_weight", 0.0):
            raise ValueError("t2s_transformer has no aux ASR / ST decoders (t2s_transformer.py:279-371)")
        args.text_encoder = True
        args.src_vocab_size = len(task.source_dictionary)  # T2STransformerEncoder(args, task.src_dict, ...)
        return super().build_model(args, task)

    def forward(self, src_tokens, src_lengths, prev_output_tokens, **kwargs):
        """t2s_transformer.py (FairseqEncoderDecoderModel.forward): returns ``(post_feat_out, eos_out, extra)``."""
        sample = {"net_input": {"prev_output_tokens": prev_output_tokens}, "src_text": src_tokens,
                  "src_text_len": src_lengths, "target_lengths": kwargs["target_lengths"],
                  "ntokens": int(kwargs["target_lengths"].sum()), "speaker": kwargs.get("speaker")}
        o = self.engine.forward(sample, training=self.training, want_attn=True, with_loss=False)
        return o["post_feat_out"], o["eos_out"], {"attn": o.get("attn"), "feature_out": o["feature_out"]}

    def forward_encoder(self, src_tokens, src_lengths, speaker=None, **kwargs):
        import torch
        B = src_tokens.shape[0]
        sample = {"net_input": {"prev_output_tokens": torch.zeros(B, 1, self.engine.cfg.out_dim)}, "src_text": src_tokens,
                  "src_text_len": src_lengths, "target_lengths": torch.ones(B, dtype=torch.long), "ntokens": B,
                  "speaker": speaker}
        o = self.engine.forward(sample, training=self.training, want_attn=False, with_loss=False)
        lens = o["encoder_lens"].long()
        E = o["encoder_out"].shape[1]
        pad = torch.arange(E, device=lens.device).unsqueeze(0) >= lens.unsqueeze(1)
        return {"encoder_out": [o["encoder_out"].transpose(0, 1)], "encoder_padding_mask": [pad] if bool(pad.any()) else [],
                "encoder_embedding": [], "encoder_states": [], "src_tokens": [], "src_lengths": []}


@register_model_architecture("t2s_transformer", "t2s_transformer")
def t2s_architecture(args):
    """Defaults of t2s_transformer.py:339-371 where they differ from ``s2st_transformer``."""
    def g(k, v):
        if getattr(args, k, None) is None:
            setattr(args, k, v)

    g("encoder_dropout", 0.5)
    g("encoder_conv_layers", 3)
    g("encoder_conv_kernel_size", 5)
    g("encoder_transformer_layers", 6)
    if not hasattr(args, "encoder_normalize_before"):
        args.encoder_normalize_before = False
    g("attention_dropout", 0.0)
    g("activation_dropout", 0.0)
    return base_architecture(args)
