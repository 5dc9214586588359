"""Train-step driver: the counterpart of ``Trainer.train_step`` (fairseq/trainer.py:709-1010)
for the MI355X engine.

Per update: seed = seed + num_updates (:1254-1258); zero grads; for each micro-batch
forward + backward (gradients accumulate in the flat arena); SUM all-reduce of gradients
(overlapped with backward) and of the sample sizes; ``grads *= world / sum(sample_size)``
(:838-843), clip by global norm (:850, fairseq/utils.py:345-395), fairseq-Adam
(fairseq/optim/adam.py:163-239) with lr = inverse_sqrt(num_updates)
(inverse_square_root_schedule.py:52-85) -- scale, clip and Adam are one HIP kernel over the
arena.  A non-finite gradient norm leaves parameters and moments untouched and bumps a device-side
counter; ``check_overflow()`` (called by the train harness at its logging interval and before every
checkpoint) then raises FloatingPointError like the reference (:860-867) -- one 4-byte read instead of a
host sync per step.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from .runtime import binding as bd
from .runtime.distributed import GradReducer, all_reduce_scalars, broadcast_, is_dist, world_size
from .runtime.engine import STAT


def inverse_sqrt_lr(num_updates: int, lr: float, warmup_updates: int, warmup_init_lr: float = -1.0) -> float:
    if warmup_init_lr < 0:
        warmup_init_lr = 0.0 if warmup_updates > 0 else lr
    if num_updates < warmup_updates:
        return warmup_init_lr + num_updates * (lr - warmup_init_lr) / warmup_updates
    return lr * warmup_updates ** 0.5 * num_updates ** -0.5


class Trainer:
    def __init__(self, args, task, model, criterion):
        self.args, self.task, self.model, self.criterion = args, task, model, criterion
        self.engine = model.engine
        dev = self.engine.device
        n = self.engine.n_params
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._sumsq_nparts = int(bd._bind("s2st_sumsq_parts_count")(self.engine.n_params))
        self.sumsq_parts = torch.zeros(max(self._sumsq_nparts, 1), dtype=torch.float32, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.gmul_dev = torch.ones(1, dtype=torch.float32, device=dev)
        self.skipped = torch.zeros(1, dtype=torch.int32, device=dev)
        self._grads_clean = False
        self.num_updates = 0
        self.lr = getattr(args, "lr", 1.5e-3)
        self.lr = self.lr[0] if isinstance(self.lr, (list, tuple)) else self.lr
        self.warmup = getattr(args, "warmup_updates", 4000)
        self.clip_norm = getattr(args, "clip_norm", 0.0)
        self.betas = tuple(getattr(args, "adam_betas", (0.9, 0.999)))
        self.eps = getattr(args, "adam_eps", 1e-8)
        self.wd = getattr(args, "weight_decay", 0.0)
        self.seed = getattr(args, "seed", 1)
        self._dummy_batch = None
        self.reducer = GradReducer(self.engine.grads, extra_stream=self.engine.side_stream(),
                                   exchange_dtype=str(getattr(args, "grad_exchange_dtype", "fp32") or "fp32")) \
            if (is_dist() or (os.environ.get("S2ST_EXCHANGE_PROXY") and dev.type == "cuda")) else None
        if is_dist():
            # DDP's constructor broadcast of parameters and buffers from rank 0
            broadcast_(self.engine.params, 0)
            broadcast_(self.engine.buffers, 0)

    def _ph(self):
        """bf16 parameter copy the optimizer kernel refreshes with the update (S2ST_ADAM_NO_PH=1: A/B switch,
        the next forward then makes the copy itself)."""
        import os
        return None if os.environ.get("S2ST_ADAM_NO_PH") else self.engine.params_bf16

    def get_lr(self) -> float:
        return inverse_sqrt_lr(self.num_updates, self.lr, self.warmup)

    def wait_optimizer(self):
        """After ``train_step(..., overlap_optimizer=True)``: make the current stream wait for the parameter update
        before anything but the engine's own forward reads the parameters (checkpoints, host copies, broadcasts)."""
        self.engine.wait_optimizer()

    def train_step(self, samples: List[Dict], fast: bool = True, overlap_optimizer: bool = False):
        """One optimizer update over ``samples`` (the update_freq micro-batches of this rank).
        ``fast`` bypasses autograd (engine.backward is called directly); ``fast=False`` goes
        through ``task.train_step`` / ``loss.backward()`` exactly like fairseq would.
        ``overlap_optimizer``: the caller's next call is another ``train_step`` (or an engine forward): the Adam kernel
        runs in chunks on the engine's second stream and the next forward waits chunk by chunk (the update hides behind
        the forward's first layers); any other reader of the parameters calls ``wait_optimizer()`` first."""
        eng = self.engine
        self.model.train()
        eng.step_seed = (self.seed + self.num_updates) * 1000003
        if not self._grads_clean:  # the previous update's Adam kernel left the arena zeroed (zero_grad = 1 below)
            eng.zero_grad()
        self._grads_clean = False
        sample_size = 0
        logs = []
        hooks = self.reducer.on_segment if self.reducer is not None else None
        for i, sample in enumerate(samples):
            last = i == len(samples) - 1
            seg_hooks = hooks if last else None  # like no_sync(): reduce once, on the last micro-batch
            # a rank whose shard ran out gets an empty batch (ShardedIterator fill value): it still has to take part
            # in the gradient exchange, so it runs the first batch it ever saw with zero weight
            # (trainer.py:1126-1160 _prepare_sample / ignore_grad, :786-789 sample_size *= 0)
            is_dummy = sample is None or len(sample) == 0
            if is_dummy:
                if self._dummy_batch is None:
                    raise RuntimeError("empty batch before any real batch was seen on this rank")
                sample = self._dummy_batch
            elif self._dummy_batch is None:
                self._dummy_batch = sample
            if fast:
                loss, ss, log = self._fast_micro_step(sample, seg_hooks, 0.0 if is_dummy else 1.0)
            else:
                self.criterion.grad_hooks = seg_hooks
                loss, ss, log = self.task.train_step(sample, self.model, self.criterion, None, self.num_updates,
                                                     ignore_grad=is_dummy)
            if not is_dummy:
                sample_size += ss
                logs.append(log)
        self._bump_bn_counters(len(samples))
        gmul_dev = None
        world = world_size()
        if self.reducer is not None:
            self.reducer.finish()
            self.gmul_dev.fill_(float(sample_size))
            all_reduce_scalars(self.gmul_dev)  # sum of sample sizes over ranks
            # DDP averages gradients and the trainer multiplies by world / sum(sample_size); the
            # arena holds the SUM over ranks, so the net factor is 1 / sum(sample_size)
            self.gmul_dev.reciprocal_()
            gmul_dev, gmul = self.gmul_dev, 1.0
        else:
            if sample_size <= 0:
                raise RuntimeError("no real batch in this update")
            gmul = 1.0 / float(sample_size)
        # gradient norm: per-block partial sums folded in index order inside the Adam kernel (no zeroing pass, no atomics:
        # the clip coefficient, hence the whole update, repeats bit for bit)
        nparts = self._sumsq_nparts
        bd.call("s2st_sumsq_parts_f32", eng.grads, eng.n_params, self.sumsq_parts)
        lr = self.get_lr()
        import os
        if overlap_optimizer:
            eng.adam_overlapped(self.exp_avg, self.exp_avg_sq, self.sumsq_parts, nparts, gmul, gmul_dev, float(self.clip_norm),
                                lr, self.betas[0], self.betas[1], self.eps, self.wd, self.num_updates + 1, self.gnorm,
                                self.skipped, self._ph() is not None, int(os.environ.get("S2ST_ADAM_CHUNKS", "16")))
        else:
            bd.call("s2st_adam_f32", eng.params, eng.grads, self.exp_avg, self.exp_avg_sq, eng.n_params,
                    self.sumsq_parts, gmul, gmul_dev, float(self.clip_norm), lr, self.betas[0], self.betas[1],
                    self.eps, self.wd, self.num_updates + 1, self.gnorm, self._ph(), self.skipped, nparts, 1)
        self._grads_clean = True
        if self._ph() is not None:
            eng.mark_bf16_fresh()
        self.num_updates += 1
        self.model.set_num_updates(self.num_updates)
        return {"logs": logs, "sample_size": sample_size, "lr": lr, "gnorm": self.gnorm}

    def check_overflow(self):
        """Raise FloatingPointError if any update since the last check met a non-finite gradient norm
        (fairseq/trainer.py:860-867 raises in the step itself; here the step stays asynchronous and the
        check is one small D2H read).  Such updates were not applied."""
        self.engine.wait_optimizer()  # (an overlapped update writes the counter from the second stream)
        n = int(self.skipped.item())
        if n:
            self.skipped.zero_()
            raise FloatingPointError(f"gradients are Nan/Inf in {n} update(s) since the last check "
                                     f"(num_updates={self.num_updates}); those updates were skipped")

    def _fast_micro_step(self, sample, hooks, gscale: float = 1.0):
        fs = getattr(self.criterion, "fast_step", None)
        if fs is not None:  # a criterion with its own sample size / logging output (s2t_loss: text tokens, not mel frames)
            return fs(self.model, sample, hooks, gscale)
        eng = self.engine
        sample = self.model.front_end_sample(sample)  # --use-hubert: frozen front end inside the step
        out = eng.forward(sample, training=True, want_attn=False, with_loss=True)
        self.criterion.last_outputs = out
        eng.backward(gscale, on_segment=hooks)
        from .criterions.s2st_loss import LazyLog
        c = eng.cfg
        log = LazyLog(out["stats"], {"ntokens": sample["ntokens"], "nsentences": sample["nsentences"],
                                     "sample_size": sample["ntokens"]},
                      getattr(self.criterion, "report_accuracy", False), bool(c.has_asr), bool(c.has_st))
        return out["stats"][STAT["LOSS"]], sample["ntokens"], log

    def _bump_bn_counters(self, n):
        for name, b in self.model.named_buffers():
            if name.endswith("num_batches_tracked"):
                b += n

    def valid_step(self, sample):
        # (validation reads the parameters through more than the training engine's own forward -- inference twins of the
        #  --eval-inference decode, torch-side copies: an overlapped update still in flight is waited for first)
        self.engine.wait_optimizer()
        return self.task.valid_step(sample, self.model, self.criterion)
