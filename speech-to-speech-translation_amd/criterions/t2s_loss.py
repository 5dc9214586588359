"""``--criterion t2s_loss`` (examples/s2s_trans/criterions/t2s_loss.py:93-210): the Tacotron-2 terms for the text-input
model -- ``src_tokens = sample["src_text"]``, ``src_lengths = sample["src_text_len"]`` (:110-121); logging keys
loss / l1_loss / mse_loss / eos_loss / attn_loss / ctc_loss."""
from __future__ import annotations

from ..registry import register_criterion
from .s2st_loss import LazyLog, Tacotron2Criterion


class _T2SLog(LazyLog):
    def materialize(self):
        if not self._done:
            super().materialize()
            for k in ("aux_asr_loss", "aux_st_loss"):
                dict.pop(self, k, None)
        return self


@register_criterion("t2s_loss")
class Tacotron2T2SCriterion(Tacotron2Criterion):
    def __init__(self, task, sentence_avg=False, n_frames_per_step=4, use_guided_attention_loss=False,
                 guided_attention_loss_sigma=0.4, bce_pos_weight=1.0, ctc_weight=0.0):
        # --ctc-weight > 0: CTC of the source text against log_softmax(ctc_proj(feature_out)) (t2s_loss.py:134-144)
        super().__init__(task, sentence_avg, n_frames_per_step, use_guided_attention_loss, guided_attention_loss_sigma,
                         bce_pos_weight, ctc_weight)

    @classmethod
    def build_criterion(cls, args, task):
        crit = cls(task, getattr(args, "sentence_avg", False), args.n_frames_per_step,
                   getattr(args, "use_guided_attention_loss", False), getattr(args, "guided_attention_loss_sigma", 0.4),
                   args.bce_pos_weight, getattr(args, "ctc_weight", 0.0))
        crit.eps = getattr(args, "label_smoothing", 0.0)
        return crit

    def forward(self, model, sample, reduction="mean"):
        loss, sample_size, log = super().forward(model, sample, reduction)
        return loss, sample_size, _T2SLog(log._stats, {k: dict.__getitem__(log, k) for k in ("ntokens", "nsentences", "sample_size")},
                                          False, False, False)
