"""``--criterion s2st_loss`` on the MI355X engine.

Mirror of examples/s2s_trans/criterions/s2st_loss.py:147-413: same constructor arguments,
``forward(model, sample, reduction="mean") -> (loss, sample_size, logging_output)``, same
logging keys, ``reduce_metrics`` and ``logging_outputs_can_be_summed() == False``.  The loss
terms and their gradients are computed by the HIP kernels (mel L1/MSE/BCE, CTC,
label-smoothed CE); ``loss.backward()`` runs the engine's backward through one autograd node.
"""
from __future__ import annotations

from typing import Any, Dict, List

import torch

from ..registry import CriterionBase, HAVE_FAIRSEQ, register_criterion
from ..runtime.engine import STAT


def label_smoothed_nll_loss(lprobs, target, epsilon, ignore_index=None, reduce=True):
    """Host restatement kept for API parity (s2st_loss.py:33-50); the training path uses the
    fused HIP kernel ``s2st_ls_ce_f32``."""
    if target.dim() == lprobs.dim() - 1:
        target = target.unsqueeze(-1)
    nll = -lprobs.gather(dim=-1, index=target)
    smooth = -lprobs.sum(dim=-1, keepdim=True)
    if ignore_index is not None:
        pm = target.eq(ignore_index)
        nll = nll.masked_fill(pm, 0.0)
        smooth = smooth.masked_fill(pm, 0.0)
    if reduce:
        nll, smooth = nll.sum(), smooth.sum()
    eps_i = epsilon / (lprobs.size(-1) - 1)
    return (1.0 - epsilon - eps_i) * nll + eps_i * smooth, nll


class _EngineStep(torch.autograd.Function):
    """One node for the whole model + loss: forward already ran on the engine; backward runs
    the engine tape, which writes parameter gradients in place into ``p.grad`` (views of the
    flat gradient arena)."""

    @staticmethod
    def forward(ctx, anchor, loss_value, engine, hooks, model=None):
        ctx.engine = engine
        ctx.hooks = hooks
        ctx.model = model
        return loss_value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        gs = float(grad_out)  # host read; the native trainer calls engine.backward directly
        if ctx.model is not None:
            attach_grad_views(ctx.model)
        ctx.engine.backward(gs, on_segment=ctx.hooks)
        return None, None, None, None, None


def attach_grad_views(model):
    """Make ``p.grad`` of every trainable parameter the view of the engine's flat gradient arena again.

    fairseq's optimizer wrapper clears gradients by DROPPING them -- ``FairseqOptimizer.zero_grad`` sets ``p.grad = None``
    (fairseq/optim/fairseq_optimizer.py:127-131), and so does ``torch.optim.Optimizer.zero_grad(set_to_none=True)`` -- while
    the engine's backward ADDS into the arena whatever the Python attributes say.  So, right before a backward: a parameter
    whose ``.grad`` is None gets its arena range zeroed (that is what "cleared" means to the caller) and its view attached;
    a parameter that still holds its view keeps accumulating (update-freq > 1: the micro-batches after the first one).
    Found by tests/test_reference_loop.py: without this the reference's own loop left the parameters where they were."""
    views = getattr(model, "_grad_views", None)
    if views is None:
        views = model._grad_views = {n: gv for n, pv, gv, isb in model.engine.named_views() if not isb}
    params = dict(model.named_parameters())
    missing = [n for n, p in params.items() if p.requires_grad and p.grad is None and n in views]
    if not missing:
        return
    if len(missing) == len(views):
        model.engine.zero_grad()  # one pass over the arena
    else:
        for n in missing:
            views[n].zero_()
    for n in missing:
        params[n].grad = views[n]


class LazyLog(dict):
    """logging_output whose numbers live on the device until somebody reads them: one D2H
    copy for all keys instead of the reference's ten ``utils.item`` syncs (s2st_loss.py:271-291)."""

    def __init__(self, stats: torch.Tensor, static: Dict[str, Any], report_accuracy, has_asr, has_st):
        super().__init__(static)
        self._stats, self._done = stats, False
        self._acc = (report_accuracy and has_asr, report_accuracy and has_st)

    def materialize(self):
        if self._done:
            return self
        s = self._stats.detach().cpu().tolist()
        for k, i in (("loss", "LOSS"), ("l1_loss", "L1"), ("mse_loss", "MSE"), ("eos_loss", "EOS"),
                     ("ctc_loss", "CTC"), ("aux_asr_loss", "ASR"), ("aux_st_loss", "ST")):
            dict.__setitem__(self, k, s[STAT[i]])
        dict.__setitem__(self, "attn_loss", 0.0)
        if self._acc[0]:
            dict.__setitem__(self, "asr_n_correct", int(s[STAT["ASR_CORRECT"]]))
            dict.__setitem__(self, "asr_total", int(s[STAT["ASR_TOTAL"]]))
        if self._acc[1]:
            dict.__setitem__(self, "st_n_correct", int(s[STAT["ST_CORRECT"]]))
            dict.__setitem__(self, "st_total", int(s[STAT["ST_TOTAL"]]))
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


@register_criterion("s2st_loss")
class Tacotron2Criterion(CriterionBase):  # fairseq's FairseqCriterion when fairseq is importable
    def __init__(self, task, sentence_avg=False, n_frames_per_step=4, use_guided_attention_loss=False,
                 guided_attention_loss_sigma=0.4, bce_pos_weight=1.0, ctc_weight=0.0, asr_ce_weight=0.0,
                 st_ce_weight=0.0, l1_loss_weight=1.0, mse_loss_weight=1.0, eos_loss_weight=1.0,
                 attn_loss_weight=1.0, label_smoothing=0.0, ignore_prefix_size=0, report_accuracy=False):
        if HAVE_FAIRSEQ:
            super().__init__(task)
        else:
            super().__init__()
        if use_guided_attention_loss:
            # the reference raises a shape error with this flag on (fbank lengths are passed to a
            # [B, E, D] attention map, s2st_loss.py:227); nothing to be compatible with
            raise NotImplementedError("guided attention loss is unusable in the reference; not built")
        self.task = task
        self.sentence_avg = sentence_avg
        self.n_frames_per_step = n_frames_per_step
        self.bce_pos_weight = bce_pos_weight
        self.ctc_weight, self.asr_ce_weight, self.st_ce_weight = ctc_weight, asr_ce_weight, st_ce_weight
        self.l1_loss_weight, self.mse_loss_weight, self.eos_loss_weight = l1_loss_weight, mse_loss_weight, eos_loss_weight
        self.eps = label_smoothing
        self.report_accuracy = report_accuracy
        self.padding_idx = 1
        self.grad_hooks = None  # set by the trainer: on_segment callback for overlapped all-reduce
        self.last_outputs = None

    @classmethod
    def build_criterion(cls, args, task):
        return cls(task, getattr(args, "sentence_avg", False), args.n_frames_per_step,
                   getattr(args, "use_guided_attention_loss", False),
                   getattr(args, "guided_attention_loss_sigma", 0.4), args.bce_pos_weight,
                   args.ctc_weight, args.asr_ce_weight, args.st_ce_weight, args.l1_loss_weight,
                   args.mse_loss_weight, args.eos_loss_weight, getattr(args, "attn_loss_weight", 1.0),
                   args.label_smoothing, report_accuracy=getattr(args, "report_accuracy", False))

    def forward(self, model, sample, reduction="mean"):
        assert reduction == "mean"
        eng = model.engine
        c = eng.cfg
        # loss weights live in the engine config (built from the same flags); keep them in sync
        c_w = (c.ctc_weight, c.asr_weight, c.st_weight, c.w_l1, c.w_mse, c.w_eos, c.bce_pos_weight, c.label_smoothing)
        mine = (self.ctc_weight, self.asr_ce_weight, self.st_ce_weight, self.l1_loss_weight,
                self.mse_loss_weight, self.eos_loss_weight, self.bce_pos_weight, self.eps)
        assert all(abs(a - b) < 1e-6 for a, b in zip(c_w, mine)), "criterion and model flags disagree"
        sample = model.front_end_sample(sample)  # --use-hubert (s2st_transformer.py:245-252)
        out = eng.forward(sample, training=model.training, want_attn=False, with_loss=True)
        self.last_outputs = out
        stats = out["stats"]
        loss_val = stats[STAT["LOSS"]]
        anchor = next(model.parameters())
        loss = _EngineStep.apply(anchor, loss_val, eng, self.grad_hooks, model) if torch.is_grad_enabled() else loss_val
        sample_size = sample["nsentences"] if self.sentence_avg else sample["ntokens"]
        log = LazyLog(stats, {"ntokens": sample["ntokens"], "nsentences": sample["nsentences"],
                              "sample_size": sample_size}, self.report_accuracy, bool(c.has_asr), bool(c.has_st))
        return loss, sample_size, log

    @classmethod
    def reduce_metrics(cls, logging_outputs: List[Dict[str, Any]]) -> Dict[str, float]:
        """Sample-size weighted means of the loss terms + accuracies (s2st_loss.py:350-407).
        Returns the scalars (fairseq's ``metrics.log_scalar`` is called when available)."""
        ns = [log.get("sample_size", 0) for log in logging_outputs]
        ntot = sum(ns)
        ws = [n / (ntot + 1e-8) for n in ns]
        res = {}
        for key in ["loss", "l1_loss", "mse_loss", "eos_loss", "attn_loss", "ctc_loss", "aux_asr_loss", "aux_st_loss"]:
            res[key] = sum(log.get(key, 0) * w for log, w in zip(logging_outputs, ws))
        res["sample_size"] = ntot
        for t in ("asr", "st"):
            tot = sum(log.get(f"{t}_total", 0) for log in logging_outputs)
            if tot > 0:
                cor = sum(log.get(f"{t}_n_correct", 0) for log in logging_outputs)
                res[f"{t}_total"], res[f"{t}_n_correct"] = tot, cor
                res[f"{t}_accuracy"] = round(cor * 100.0 / tot, 3)
        # inference metrics (--eval-inference: s2st_loss.py:394-407)
        if logging_outputs and "targ_frames" in logging_outputs[0]:
            n = sum(log.get("targ_frames", 0) for log in logging_outputs)
            for key, new_key in [("mcd_loss", "mcd_loss"), ("pred_frames", "pred_ratio"), ("nins", "ins_rate"),
                                 ("ndel", "del_rate")]:
                res[new_key] = sum(log.get(key, 0) for log in logging_outputs) / n
        try:  # pragma: no cover
            from fairseq import metrics
            for k, v in res.items():
                metrics.log_scalar(k, v, ntot if k != "sample_size" else 1, round=3)
        except Exception:
            pass
        return res

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return False
