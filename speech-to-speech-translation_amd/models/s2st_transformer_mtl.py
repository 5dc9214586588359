"""``--arch s2st_transformer_mtl`` on the MI355X engine.

Host-side mirror of examples/s2s_trans/models/s2st_transformer_mtl.py:410-638: the speech encoder + mel decoder of
``s2st_transformer`` without the aux text decoders, with the encoder-tap CTC head (source text) and -- the variant's
addition -- a second CTC head ``decoder.ctc_proj_tgt`` over the TARGET text on the raw output of decoder layer
``--middle-layers-decoder`` (:266-271, 325-327, 366-371).  Same engine, same kernels: ``has_ctc_tgt`` / ``tap_dec``
switch the head on (include/s2st_hip.h: s2st_model_config).  Its architecture function differs in two defaults
(encoder width 256 with an 8x feed-forward, :610-612).
"""
from __future__ import annotations

import torch

from ..registry import register_model, register_model_architecture
from .s2st_transformer import S2STTransformerModel, base_architecture


@register_model("s2st_transformer_mtl")
class S2STTransformerMTLModel(S2STTransformerModel):
    @staticmethod
    def add_args(parser):
        S2STTransformerModel.add_args(parser)
        parser.add_argument("--middle-layers-decoder", default="6", type=str)

    @classmethod
    def build_model(cls, args, task):
        mtl_architecture(args)
        if getattr(args, "asr_ce_weight", 0.0) or getattr(args, "st_ce_weight", 0.0):
            raise ValueError("s2st_transformer_mtl has no aux ASR / ST decoders (s2st_transformer_mtl.py:410-560)")
        if str(getattr(args, "use_hubert", "false")) == "true":
            raise NotImplementedError("the mtl variant's encoder has no HuBERT branch (s2st_transformer_mtl.py:172-175)")
        return super().build_model(args, task)

    def get_normalized_probs(self, net_output, log_probs, sample=None, tag="ctc"):
        """s2st_transformer_mtl.py:366-374: ``tag="ctc"`` -> source-text head over the encoder tap, ``"ctc_tgt"`` ->
        target-text head over the decoder tap (``net_output[2]`` is then the list holding that tap, [D, B, C])."""
        if tag != "ctc_tgt":
            return super().get_normalized_probs(net_output, log_probs, sample)
        if not self.engine.cfg.has_ctc_tgt:
            raise ValueError("the model was built without the target-text CTC head (--ctc-weight-tgt 0)")
        from ..runtime import binding as bd
        taps = net_output[2]
        tap = (taps["out_middle_layers_decoder"] if isinstance(taps, dict) else taps)[0].transpose(0, 1).contiguous()
        B, D, Cd = tap.shape
        w, b = self._views["decoder.ctc_proj_tgt.weight"], self._views["decoder.ctc_proj_tgt.bias"]
        V = w.shape[0]
        ld = (V + 3) // 4 * 4
        logits = torch.empty(B * D, ld, dtype=torch.float32, device=tap.device)
        bd.gemm(tap.view(B * D, Cd), w, logits, B * D, V, Cd, c_ld=ld, bias=b, precise=True)
        out = torch.empty(B * D, V, dtype=torch.float32, device=tap.device)
        bd.call("s2st_log_softmax_rows_f32", logits, ld, out, V, B * D, V, 1 if log_probs else 0)
        return out.view(B, D, V)


@register_model_architecture("s2st_transformer_mtl", "s2st_transformer_mtl")
def mtl_architecture(args):
    """Defaults of s2st_transformer_mtl.py:604-638 where they differ from ``s2st_transformer``."""
    def g(k, v):
        if getattr(args, k, None) is None:
            setattr(args, k, v)

    g("encoder_embed_dim", 256)
    g("encoder_ffn_embed_dim", 8 * args.encoder_embed_dim)
    g("middle_layers_decoder", "6")
    g("ctc_weight_tgt", 0.0)
    return base_architecture(args)
