"""Checkpoints in the reference's on-disk format.

Counterpart of ``fairseq/trainer.py:380-424`` (``Trainer.state_dict``), ``:438-560`` (``load_checkpoint``) and
``fairseq/checkpoint_utils.py:281-345, 513-541`` for the s2s_translation path: a ``.pt`` file written by the reference
(``model`` state dict with the names of SURVEY Appendix A, ``optimizer_history``, ``last_optimizer_state`` of
torch's ``Optimizer.state_dict()`` over ``model.parameters()`` order -- or fairseq's flattened fp32 master copy when
it trained with ``--fp16``) resumes here, and a file written here resumes there: same keys, same per-parameter
Adam state, ``cfg["model"]`` kept as a namespace (what ``examples/s2s_trans/convert_pt_to512.py`` patches).
The optimizer state lives in the engine's flat arenas; it is scattered / gathered by parameter name.
"""
from __future__ import annotations

import argparse
import os
from typing import Any, Dict, Optional

import torch


def load_checkpoint_to_cpu(path: str) -> Dict[str, Any]:
    if not os.path.isfile(path):
        raise FileNotFoundError(f"Model file not found: {path}")
    # the reference pickles argparse.Namespace / plain containers next to the tensors
    state = torch.load(path, map_location="cpu", weights_only=False)
    if "optimizer_history" not in state:  # checkpoint_utils._upgrade_state_dict, oldest layouts
        state["optimizer_history"] = [{"criterion_name": "Tacotron2Criterion", "optimizer_name": "FairseqAdam",
                                       "lr_scheduler_state": {"best": None}, "num_updates": 0}]
    return state


def load_pretrained_component_from_model(model, component_type: str, checkpoint: str):
    """fairseq/checkpoint_utils.py:784-812 for ``--load-pretrained-encoder-from`` / ``--load-pretrained-decoder-from``
    (examples/s2s_trans/models/s2st_transformer.py:704-733): the ``encoder`` / ``decoder`` entries of the checkpoint's
    model state are loaded into the module of that name, non-strictly like the reference (entries the module does not
    have, or lacks, are skipped; a shape mismatch is an error, as in ``load_state_dict``).  The parameters are views of
    the engine's arena, so the copy lands in place."""
    if component_type not in ("encoder", "decoder"):
        raise ValueError("component to load must be either the encoder or the decoder")
    state = load_checkpoint_to_cpu(checkpoint)
    comp = getattr(model, component_type)
    sub = {k[len(component_type) + 1:]: v for k, v in state["model"].items() if k.startswith(component_type + ".")}
    res = comp.load_state_dict(sub, strict=False)
    # (the copies bump the arena tensor's version: the engine's next forward refreshes its bf16 operand copy)
    return [k for k in sub if k not in res.unexpected_keys]


def _arena_slices(engine):
    """name -> (offset, numel) of every trainable parameter inside the flat arenas."""
    base = engine.params.data_ptr()
    return {n: ((pv.data_ptr() - base) // 4, pv.numel()) for n, pv, gv, isb in engine.named_views() if not isb}


def load_optimizer_state(trainer, opt_state: Dict[str, Any]) -> int:
    """Scatter a torch/fairseq Adam ``state_dict()`` into the trainer's moment arenas; returns the step count."""
    model, eng = trainer.model, trainer.engine
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    where = _arena_slices(eng)
    st = opt_state["state"]
    total = sum(where[n][1] for n in names)
    steps = set()
    trainer.exp_avg.zero_()
    trainer.exp_avg_sq.zero_()
    if len(st) == 1 and next(iter(st.values()))["exp_avg"].numel() == total and len(names) > 1:
        # --fp16: one flattened fp32 master parameter (fairseq/optim/fp16_optimizer.py:77-95), model.parameters() order
        s0 = next(iter(st.values()))
        steps.add(int(s0["step"]))
        o = 0
        for n in names:
            off, k = where[n]
            trainer.exp_avg[off:off + k].copy_(s0["exp_avg"][o:o + k].float())
            trainer.exp_avg_sq[off:off + k].copy_(s0["exp_avg_sq"][o:o + k].float())
            o += k
    else:
        ids = opt_state["param_groups"][0]["params"] if opt_state.get("param_groups") else sorted(st.keys())
        if len(ids) != len(names):
            raise ValueError(f"optimizer state covers {len(ids)} parameters, the model has {len(names)}")
        for pid, n in zip(ids, names):
            if pid not in st:  # parameter that never received a gradient
                continue
            off, k = where[n]
            s = st[pid]
            if s["exp_avg"].numel() != k:
                raise ValueError(f"optimizer state of {n}: {s['exp_avg'].numel()} elements, parameter has {k}")
            steps.add(int(s["step"]))
            trainer.exp_avg[off:off + k].copy_(s["exp_avg"].reshape(-1).float())
            trainer.exp_avg_sq[off:off + k].copy_(s["exp_avg_sq"].reshape(-1).float())
    if len(steps) > 1:
        raise ValueError(f"per-parameter Adam step counts differ ({sorted(steps)}): the fused optimizer keeps one")
    return steps.pop() if steps else 0


def optimizer_state_dict(trainer) -> Dict[str, Any]:
    """Gather the moment arenas into torch's ``Optimizer.state_dict()`` layout over model.parameters() order."""
    names = [n for n, p in trainer.model.named_parameters() if p.requires_grad]
    where = _arena_slices(trainer.engine)
    shapes = dict((n, tuple(p.shape)) for n, p in trainer.model.named_parameters())
    state = {}
    if trainer.num_updates > 0:
        for i, n in enumerate(names):
            off, k = where[n]
            state[i] = {"step": trainer.num_updates,
                        "exp_avg": trainer.exp_avg[off:off + k].detach().cpu().clone().view(shapes[n]),
                        "exp_avg_sq": trainer.exp_avg_sq[off:off + k].detach().cpu().clone().view(shapes[n])}
    group = {"lr": trainer.get_lr(), "betas": tuple(trainer.betas), "eps": trainer.eps, "weight_decay": trainer.wd,
             "amsgrad": False, "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}


def state_dict(trainer, extra_state: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    args = trainer.args
    ns = argparse.Namespace(**vars(args)) if not isinstance(args, dict) else argparse.Namespace(**args)
    return {
        "args": None,  # legacy slot, as in the reference
        "cfg": {"model": ns, "task": {"_name": "s2s_translation"}, "criterion": {"_name": "s2st_loss"},
                "optimizer": {"_name": "adam"}, "lr_scheduler": {"_name": "inverse_sqrt"}},
        "model": {k: v.detach().cpu().clone() for k, v in trainer.model.state_dict().items()},
        "criterion": None,
        "optimizer_history": [{"criterion_name": "Tacotron2Criterion", "optimizer_name": "FairseqAdam",
                               "lr_scheduler_state": {"best": getattr(trainer, "best", None)},
                               "num_updates": trainer.num_updates}],
        "task_state": {},
        # (the reference's loader upgrades files without a stateful "train_iterator" entry from legacy keys)
        "extra_state": dict({"train_iterator": {"epoch": 1, "iterations_in_epoch": 0}, "previous_training_time": 0},
                            **(extra_state or {})),
        "last_optimizer_state": optimizer_state_dict(trainer),
    }


def save_checkpoint(path: str, trainer, extra_state: Optional[Dict[str, Any]] = None) -> None:
    tmp = path + ".tmp"  # atomic like torch_persistent_save
    if hasattr(trainer, "wait_optimizer"):
        trainer.wait_optimizer()  # (an update overlapped with the next forward may still be in flight on the second stream)
    torch.save(state_dict(trainer, extra_state), tmp)
    os.replace(tmp, path)


def load_checkpoint(path: str, trainer, reset_optimizer: bool = False, reset_lr_scheduler: bool = False) -> Dict[str, Any]:
    """Model (strict), optimizer moments + step, update counter; returns ``extra_state``."""
    state = load_checkpoint_to_cpu(path)
    trainer.model.load_state_dict(state["model"], strict=True)
    last = state["optimizer_history"][-1]
    opt_state = state.get("last_optimizer_state")
    if opt_state is not None and not reset_optimizer:
        if last["criterion_name"] != "Tacotron2Criterion":
            raise ValueError(f"Criterion does not match; please reset the optimizer ({last['criterion_name']})")
        if last["optimizer_name"] not in ("FairseqAdam", "MemoryEfficientFP16Optimizer", "FP16Optimizer"):
            raise ValueError(f"Optimizer does not match; please reset the optimizer ({last['optimizer_name']})")
        step = load_optimizer_state(trainer, opt_state)
        if not reset_lr_scheduler:
            trainer.best = last["lr_scheduler_state"].get("best")
        trainer.num_updates = int(last["num_updates"])
        if step not in (0, trainer.num_updates):
            raise ValueError(f"Adam step count {step} != num_updates {trainer.num_updates}")
        trainer.model.set_num_updates(trainer.num_updates)
    return state.get("extra_state") or {}
