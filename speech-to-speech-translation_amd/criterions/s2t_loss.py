"""``--criterion s2t_loss`` (examples/s2s_trans/criterions/s2t_loss.py:57-198): label-smoothed cross entropy of the
``s2t_transformer_hubert`` text decoder, ``--test-type asr`` (source text) or ``st`` (target text).

Same constructor arguments, ``forward(model, sample, reduce=True) -> (loss, sample_size, logging_output)`` with the keys
loss / nll_loss / ntokens / nsentences / sample_size (+ n_correct / total with ``--report-accuracy``), ``reduce_metrics``
(loss and nll_loss in bits, ppl, accuracy) and ``logging_outputs_can_be_summed() == True``.  Loss, its gradient and the
accuracy counts come from the fused HIP kernel ``s2st_ls_ce_f32`` inside the engine's step (summed over the non-pad
tokens: ``reduce=True``, the only form the trainer uses)."""
from __future__ import annotations

import math
from typing import Any, Dict, List

import torch

from ..registry import CriterionBase, HAVE_FAIRSEQ, register_criterion
from ..runtime.engine import STAT
from .s2st_loss import _EngineStep


class S2TLog(dict):
    """logging_output whose numbers stay on the device until read (one copy for all keys)."""

    def __init__(self, stats: torch.Tensor, static: Dict[str, Any], report_accuracy: bool):
        super().__init__(static)
        self._stats, self._done, self._acc = stats, False, report_accuracy

    def materialize(self):
        if not self._done:
            s = self._stats.detach().cpu().tolist()
            dict.__setitem__(self, "loss", s[STAT["LOSS"]])
            dict.__setitem__(self, "nll_loss", s[STAT["ASR_NLL"]])
            if self._acc:
                dict.__setitem__(self, "n_correct", int(s[STAT["ASR_CORRECT"]]))
                dict.__setitem__(self, "total", int(s[STAT["ASR_TOTAL"]]))
            self._done = True
        return self

    def __getitem__(self, k):
        if not self._done and k not in ("ntokens", "nsentences", "sample_size"):
            self.materialize()
        return dict.__getitem__(self, k)

    def get(self, k, d=None):
        try:
            return self[k]
        except KeyError:
            return d

    def items(self):
        return self.materialize() and dict.items(self)

    def keys(self):
        return self.materialize() and dict.keys(self)


@register_criterion("s2t_loss")
class LabelSmoothedCrossEntropyCriterion(CriterionBase):
    def __init__(self, task, sentence_avg=False, label_smoothing=0.0, ignore_prefix_size=0, report_accuracy=False,
                 test_type="asr"):
        if HAVE_FAIRSEQ:
            super().__init__(task)
        else:
            super().__init__()
        if ignore_prefix_size:
            raise NotImplementedError("--ignore-prefix-size > 0 is not built on the HIP path (the recipes use 0)")
        if test_type not in ("asr", "st"):
            raise ValueError("--test-type must be asr or st")
        self.task, self.sentence_avg, self.eps = task, sentence_avg, label_smoothing
        self.report_accuracy, self.test_type = report_accuracy, test_type
        self.padding_idx = 1
        self.grad_hooks = None
        self.last_outputs = None

    @staticmethod
    def add_args(parser):
        """s2t_loss.py:16-35 (--label-smoothing / --report-accuracy are shared flags of the harness)."""
        parser.add_argument("--ignore-prefix-size", type=int, default=0)
        parser.add_argument("--test-type", type=str, default="asr", help="test asr or st")

    @classmethod
    def build_criterion(cls, args, task):
        return cls(task, getattr(args, "sentence_avg", False), getattr(args, "label_smoothing", 0.0),
                   getattr(args, "ignore_prefix_size", 0), getattr(args, "report_accuracy", False),
                   getattr(args, "test_type", "asr"))

    def fast_step(self, model, sample, hooks, gscale: float = 1.0):
        """The trainer's autograd-free micro step (trainer.py: _fast_micro_step): forward + loss, then the engine's backward
        directly; same return triple as ``forward``."""
        with torch.no_grad():
            loss, sample_size, log = self.forward(model, sample)
        model.engine.backward(gscale, on_segment=hooks)
        return loss, sample_size, log

    def forward(self, model, sample, reduce=True):
        assert reduce, "the engine sums the loss over the batch (reduce=True)"
        eng = model.engine
        if not eng.cfg.s2t_mode:
            raise ValueError("--criterion s2t_loss goes with --arch s2t_transformer_hubert")
        assert abs(eng.cfg.label_smoothing - self.eps) < 1e-6, "criterion and model flags disagree (--label-smoothing)"
        eng.s2t_test_type = model.test_type = self.test_type
        sample = model.front_end_sample(sample)  # --use-hubert (s2t_transformer_me.py:398-405)
        out = eng.forward(sample, training=model.training, want_attn=False, with_loss=True)
        self.last_outputs = out
        stats = out["stats"]
        loss_val = stats[STAT["LOSS"]]
        anchor = next(model.parameters())
        loss = _EngineStep.apply(anchor, loss_val, eng, self.grad_hooks, model) if torch.is_grad_enabled() else loss_val
        key = "src" if self.test_type == "asr" else "tgt"
        nsent = int(sample[f"{key}_text"].size(0))
        ntok = sample[f"{key}_txt_ntokens"]
        sample_size = nsent if self.sentence_avg else ntok
        return loss, sample_size, S2TLog(stats, {"ntokens": ntok, "nsentences": nsent, "sample_size": sample_size},
                                         self.report_accuracy)

    @classmethod
    def reduce_metrics(cls, logging_outputs: List[Dict[str, Any]]) -> Dict[str, float]:
        """s2t_loss.py:152-187: loss / sample_size and nll_loss / ntokens in bits, ppl = 2 ** nll_loss, accuracy."""
        loss_sum = sum(log.get("loss", 0) for log in logging_outputs)
        nll_sum = sum(log.get("nll_loss", 0) for log in logging_outputs)
        ntokens = sum(log.get("ntokens", 0) for log in logging_outputs)
        sample_size = sum(log.get("sample_size", 0) for log in logging_outputs)
        res = {"loss": loss_sum / sample_size / math.log(2), "nll_loss": nll_sum / ntokens / math.log(2),
               "sample_size": sample_size}
        res["ppl"] = round(2 ** res["nll_loss"], 2) if res["nll_loss"] < 128 else float("inf")  # utils.get_perplexity
        total = sum(log.get("total", 0) for log in logging_outputs)
        if total > 0:
            n_correct = sum(log.get("n_correct", 0) for log in logging_outputs)
            res["total"], res["n_correct"] = total, n_correct
            res["accuracy"] = round(n_correct * 100.0 / total, 3)
        try:  # pragma: no cover
            from fairseq import metrics
            metrics.log_scalar("loss", res["loss"], sample_size, round=3)
            metrics.log_scalar("nll_loss", res["nll_loss"], ntokens, round=3)
        except Exception:
            pass
        return res

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return True
