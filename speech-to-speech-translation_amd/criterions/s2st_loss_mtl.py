"""``--criterion s2st_loss_mtl`` (examples/s2s_trans/criterions/s2st_loss_mtl.py:100-260): the Tacotron-2 terms + CTC over
the source text on the encoder tap (``--ctc-weight``) + CTC over the target text on a decoder layer's output
(``--ctc-weight-tgt``); no aux cross-entropy terms.  Logging keys: the reference's (``ctc_loss_tgt`` added)."""
from __future__ import annotations

from typing import Any, Dict, List

from ..registry import register_criterion
from ..runtime.engine import STAT
from .s2st_loss import LazyLog, Tacotron2Criterion


class _MtlLog(LazyLog):
    def materialize(self):
        if not self._done:
            super().materialize()
            for k in ("aux_asr_loss", "aux_st_loss"):
                dict.pop(self, k, None)
            dict.__setitem__(self, "ctc_loss_tgt", float(self._stats[STAT["CTC_TGT"]]))
        return self


@register_criterion("s2st_loss_mtl")
class Tacotron2MTLCriterion(Tacotron2Criterion):
    def __init__(self, task, sentence_avg=False, n_frames_per_step=4, use_guided_attention_loss=False,
                 guided_attention_loss_sigma=0.4, bce_pos_weight=1.0, ctc_weight=0.0, ctc_weight_tgt=0.0):
        super().__init__(task, sentence_avg, n_frames_per_step, use_guided_attention_loss, guided_attention_loss_sigma,
                         bce_pos_weight, ctc_weight)
        self.ctc_weight_tgt = ctc_weight_tgt

    @staticmethod
    def add_args(parser):
        """The one field the mtl criterion's dataclass adds (s2st_loss_mtl.py:52-98 ``ctc_weight_tgt``)."""
        parser.add_argument("--ctc-weight-tgt", type=float, default=0.0,
                            help="weight of the CTC loss over the target text on a decoder layer's output")

    @classmethod
    def build_criterion(cls, args, task):
        crit = cls(task, getattr(args, "sentence_avg", False), args.n_frames_per_step,
                   getattr(args, "use_guided_attention_loss", False), getattr(args, "guided_attention_loss_sigma", 0.4),
                   args.bce_pos_weight, args.ctc_weight, getattr(args, "ctc_weight_tgt", 0.0))
        crit.eps = getattr(args, "label_smoothing", 0.0)  # unused here (no CE terms); kept equal to the model's flag
        return crit

    def forward(self, model, sample, reduction="mean"):
        assert abs(model.engine.cfg.ctc_tgt_weight - self.ctc_weight_tgt) < 1e-6, "criterion and model flags disagree"
        loss, sample_size, log = super().forward(model, sample, reduction)
        mlog = _MtlLog(log._stats, {k: dict.__getitem__(log, k) for k in ("ntokens", "nsentences", "sample_size")},
                       False, False, False)
        return loss, sample_size, mlog

    @classmethod
    def reduce_metrics(cls, logging_outputs: List[Dict[str, Any]]) -> Dict[str, float]:
        res = super().reduce_metrics(logging_outputs)
        ns = [log.get("sample_size", 0) for log in logging_outputs]
        ntot = sum(ns)
        res["ctc_loss_tgt"] = sum(log.get("ctc_loss_tgt", 0) * n / (ntot + 1e-8) for log, n in zip(logging_outputs, ns))
        for k in ("aux_asr_loss", "aux_st_loss"):
            res.pop(k, None)
        return res
