"""``train``: the counterpart of ``fairseq_cli/train.py:49-205, 236-320`` (+ the parts of
``fairseq/checkpoint_utils.py:34-187`` and ``fairseq/trainer.py`` it drives) for the MI355X path.

    python -m s2st_amd.train DATA --config-yaml config.yaml --train-subset train_fisher --valid-subset dev_fisher \
        --task s2s_translation --arch s2st_transformer --criterion s2st_loss --max-tokens 60000 --max-update 100000 \
        --n-frames-per-step 4 --bce-pos-weight 5.0 --lr 1.5e-3 --warmup-updates 4000 --clip-norm 1.0 ... --save-dir DIR

takes the flags of the reference recipe (examples/s2s_trans/run_baseline.sh:96-124) unchanged: flags -> task -> model /
criterion (through the registry) -> trainer -> [resume] -> epoch loop (sharded, per-epoch shuffled batches; update-freq
micro-batches per update; background upload of the next batches) -> validation (loss, and MCD with --eval-inference)
-> checkpoints in the reference's ``.pt`` layout.  One process per GPU: launched under ``torch.distributed.run`` it
reads RANK / WORLD_SIZE / LOCAL_RANK, shards the batches round-robin and all-reduces gradients over RCCL.
Returns a summary dict (losses, validation history) so tests can drive it in-process.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from typing import Dict, List, Optional

import numpy as np
import torch

from . import checkpoint_utils
from .registry import ARCHS, CRITERIA, MODELS, TASKS
from .runtime.prefetch import DevicePrefetcher
from .trainer import Trainer


def get_parser() -> argparse.ArgumentParser:
    """fairseq's ``options.get_training_parser`` for this path: the recipe's flags, by their fairseq names."""
    p = argparse.ArgumentParser(prog="s2st_amd.train", allow_abbrev=False)
    a = p.add_argument
    a("--user-dir", default=None, help="accepted for command-line compatibility (this package IS the plugin)")
    a("--task", default="s2s_translation", choices=sorted(TASKS) or None)
    a("--arch", "-a", default="s2st_transformer")
    a("--criterion", default="s2st_loss")
    # dataset / batching (fairseq/dataclass/configs.py DatasetConfig)
    a("--train-subset", default="train")
    a("--valid-subset", default="valid")
    a("--max-tokens", type=int, default=20000)
    a("--batch-size", "--max-sentences", type=int, default=None, dest="batch_size")
    a("--required-batch-size-multiple", type=int, default=8)
    a("--num-workers", type=int, default=1, help="loader processes that read and collate batches ahead of the trainer "
                                                 "(data/iterators.py); a further thread + HIP stream uploads them "
                                                 "(runtime/prefetch.py)")
    a("--skip-invalid-size-inputs-valid-test", action="store_true")
    a("--disable-validation", action="store_true")
    a("--validate-interval", type=int, default=1)
    a("--validate-interval-updates", type=int, default=0)
    a("--validate-after-updates", type=int, default=0)
    a("--max-tokens-valid", type=int, default=None)
    # optimization (OptimizationConfig, adam, inverse_sqrt)
    a("--max-epoch", type=int, default=0)
    a("--max-update", type=int, default=0)
    a("--clip-norm", type=float, default=0.0)
    a("--update-freq", type=int, default=1)
    a("--lr", type=float, default=1.5e-3)
    a("--optimizer", default="adam", choices=["adam"])
    a("--adam-betas", default="(0.9, 0.999)")
    a("--adam-eps", type=float, default=1e-8)
    a("--weight-decay", type=float, default=0.0)
    a("--lr-scheduler", default="inverse_sqrt", choices=["inverse_sqrt"])
    a("--warmup-updates", type=int, default=4000)
    a("--seed", type=int, default=1)
    a("--fp16", action="store_true", help="accepted: the engine's fast mode (bf16 MFMA operands, fp32 master weights / "
                                          "accumulation) is what runs; no loss scaling is needed")
    a("--precise-gemm", action="store_true", help="bf16x3 GEMMs (fp32-accurate; parity runs)")
    a("--find-unused-parameters", action="store_true", help="accepted (frozen / unused heads are simply not reduced)")
    # checkpoints (CheckpointConfig)
    a("--save-dir", default="checkpoints")
    a("--restore-file", default="checkpoint_last.pt")
    a("--reset-optimizer", action="store_true")
    a("--reset-lr-scheduler", action="store_true")
    a("--reset-meters", action="store_true")
    a("--save-interval", type=int, default=1)
    a("--save-interval-updates", type=int, default=0)
    a("--no-save", action="store_true")
    a("--keep-last-epochs", type=int, default=-1)
    a("--keep-best-checkpoints", type=int, default=-1)
    a("--best-checkpoint-metric", default="loss")
    a("--maximize-best-checkpoint-metric", action="store_true")
    # logging
    a("--log-interval", type=int, default=100)
    a("--log-format", default="json")
    a("--log-file", default=None)
    a("--tensorboard-logdir", default=None, help="accepted; scalars go to --log-file / stdout as json lines")
    # criterion (examples/s2s_trans/criterions/s2st_loss.py:52-103 Tacotron2CriterionConfig)
    a("--bce-pos-weight", type=float, default=1.0)
    a("--use-guided-attention-loss", action="store_true")
    a("--guided-attention-loss-sigma", type=float, default=0.4)
    a("--ctc-weight", type=float, default=0.0)
    a("--asr-ce-weight", type=float, default=0.0)
    a("--st-ce-weight", type=float, default=0.0)
    a("--l1-loss-weight", type=float, default=1.0)
    a("--mse-loss-weight", type=float, default=1.0)
    a("--eos-loss-weight", type=float, default=1.0)
    a("--attn-loss-weight", type=float, default=1.0)
    a("--label-smoothing", type=float, default=0.0)
    a("--report-accuracy", action="store_true")
    a("--sentence-avg", action="store_true")
    a("--spec-bwd-max-iter", type=int, default=32)
    a("--grad-exchange-dtype", default="fp32", choices=["fp32", "bf16"],
      help="type the gradient ranges are all-reduced in (fp32: the reference's DDP; bf16: half the wire bytes, one rounding "
           "of every rank's contribution -- runtime/distributed.py)")
    return p


def parse_args(argv: Optional[List[str]] = None) -> argparse.Namespace:
    """fairseq/options.py:88-219 (``parse_args_and_arch``): a first pass finds ``--task`` / ``--arch`` / ``--criterion``,
    then the flags of exactly those classes are added and the command line is parsed for good."""
    from . import criterions, models, tasks  # noqa: F401  (fill the registries)
    pre, _ = get_parser().parse_known_args(argv)
    if pre.arch not in ARCHS:
        raise SystemExit(f"unknown --arch {pre.arch}; known: {sorted(ARCHS)}")
    if pre.criterion not in CRITERIA:
        raise SystemExit(f"unknown --criterion {pre.criterion}; known: {sorted(CRITERIA)}")
    p = get_parser()
    TASKS[pre.task].add_args(p)
    MODELS[ARCHS[pre.arch][0]].add_args(p)
    crit_args = getattr(CRITERIA[pre.criterion], "add_args", None)
    if crit_args is not None:
        crit_args(p)
    args = p.parse_args(argv)
    if isinstance(args.adam_betas, str):
        args.adam_betas = tuple(float(x) for x in args.adam_betas.strip("()[] ").split(","))
    # store_true flags of the model parser that the architecture function must see as "unset"
    ARCHS[args.arch][1](args)
    return args


def _dist_init(device_index: Optional[int] = None):
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": torch.device("cuda", device_index)} if backend == "nccl" and device_index is not None else {}
        torch.distributed.init_process_group(backend, **kw)
    return world, rank


class _Log:
    def __init__(self, path, rank):
        self.f = open(path, "a") if path and rank == 0 else None
        self.rank = rank

    def __call__(self, **rec):
        if self.rank != 0:
            return
        line = json.dumps(rec)
        print(line, flush=True)
        if self.f:
            self.f.write(line + "\n")
            self.f.flush()


def validate(args, trainer: Trainer, task, subsets: List[str], world: int, rank: int) -> Dict[str, float]:
    """fairseq_cli/train.py:399-448: every batch of each valid subset through ``task.valid_step``, aggregated by the
    criterion's ``reduce_metrics``."""
    out = {}
    for subset in subsets:
        ds = task.dataset(subset) if subset in task.datasets else task.load_dataset(subset)
        itr = task.get_batch_iterator(ds, max_tokens=args.max_tokens_valid or args.max_tokens,
                                      max_sentences=args.batch_size, max_positions=task.max_positions(),
                                      required_batch_size_multiple=args.required_batch_size_multiple, seed=args.seed,
                                      num_shards=world, shard_id=rank)
        logs = []
        for sample in itr.next_epoch_itr(shuffle=False):
            if sample is None or len(sample) == 0:
                continue
            _, _, log = trainer.valid_step(sample)
            logs.append(dict(log.items()))
        if world > 1:
            gathered = [None] * world
            torch.distributed.all_gather_object(gathered, logs)
            logs = [x for part in gathered for x in part]
        res = task.reduce_metrics(logs, trainer.criterion) if logs else {}
        out.update({(k if len(subsets) == 1 else f"{subset}_{k}"): v for k, v in res.items()})
    return out


def best_checkpoint_files(existing: List[str], metric: str, keep_best: int, maximize: bool, v: float, best: Optional[float],
                          epoch: int, updates: int):
    """Which "best" files a validation score writes -- fairseq/checkpoint_utils.py:41-44, 65-104.  Returns (updated best,
    file names).  ``is_better`` counts a tie as better; ``best`` is updated FIRST and the conditions compare against the
    updated value.  ``checkpoint.best_<metric>_<v:.3f><d>.pt`` (d = a digit in [0, keep_best) drawn under
    ``numpy_seed(epoch, updates, v)`` so that equal scores do not overwrite each other) is written only when the score is
    at least as good as the WORST kept one -- whose value the reference parses back from its file name, tie-break digit
    included (reproduced) -- or, with none kept yet, as the best so far."""
    import re
    from .data.iterators import numpy_seed
    is_better = (lambda a_, b_: a_ >= b_) if maximize else (lambda a_, b_: a_ <= b_)
    best = v if best is None else (max(v, best) if maximize else min(v, best))
    names = []
    if is_better(v, best):
        names.append("checkpoint_best.pt")
    if keep_best > 0:
        rx = re.compile(r"checkpoint\.best_%s_(\d+\.?\d*)\.pt" % metric)  # (the reference does not escape the metric either)
        kept = sorted((float(m.group(1)) for m in (rx.fullmatch(fn) for fn in existing) if m), reverse=True)
        worst_best = best
        if kept:
            worst_best = kept[-1] if maximize else kept[0]
        with numpy_seed(epoch, updates, v):
            rand_sfx = np.random.randint(0, keep_best)
        if worst_best is None or is_better(v, worst_best):
            names.append("checkpoint.best_{}_{:.3f}{}.pt".format(metric, v, rand_sfx))
    return best, names


def main(argv: Optional[List[str]] = None, device: Optional[torch.device] = None, args: Optional[argparse.Namespace] = None,
         on_model_built=None) -> Dict:
    args = args or parse_args(argv)
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if device is None:
        if not torch.cuda.is_available():
            raise SystemExit("s2st_amd.train needs a HIP device (the product path has no CPU fallback)")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    world, rank = _dist_init(local_rank if device.type == "cuda" else None)
    log = _Log(args.log_file, rank)
    torch.set_num_threads(1)  # host-side torch work here is KB-sized index tensors: intra-op threads only add latency
    torch.manual_seed(args.seed)  # fairseq_cli/train.py:75-76
    task = TASKS[args.task].setup_task(args, device=device)
    if not hasattr(args, "src_vocab_size") or task.data_cfg is not None:
        args.src_vocab_size, args.tgt_vocab_size = len(task.source_dictionary), len(task.target_dictionary)
    valid_subsets = [s for s in args.valid_subset.split(",") if s] if not args.disable_validation else []
    task.load_dataset(args.train_subset)
    model = task.build_model(args)
    if on_model_built is not None:
        on_model_built(model)
    criterion = task.build_criterion(args)
    trainer = Trainer(args, task, model, criterion)
    n_par = sum(p.numel() for p in model.parameters())
    log(event="start", params=int(n_par), world_size=world, arch=args.arch, task=args.task, criterion=args.criterion)

    epoch_itr = task.get_batch_iterator(
        task.dataset(args.train_subset), max_tokens=args.max_tokens, max_sentences=args.batch_size,
        max_positions=task.max_positions(), required_batch_size_multiple=args.required_batch_size_multiple,
        seed=args.seed, num_shards=world, shard_id=rank,
        # one worker = the staging thread below (it runs the iterator itself); more = a pool of loader processes
        num_workers=args.num_workers if args.num_workers > 1 else 0)
    os.makedirs(args.save_dir, exist_ok=True)
    restore = args.restore_file if os.path.isabs(args.restore_file) else os.path.join(args.save_dir, args.restore_file)
    if os.path.isfile(restore):  # checkpoint_utils.load_checkpoint (fairseq/checkpoint_utils.py:190-278)
        extra = checkpoint_utils.load_checkpoint(restore, trainer, reset_optimizer=args.reset_optimizer,
                                                 reset_lr_scheduler=args.reset_lr_scheduler)
        if extra and extra.get("train_iterator"):
            epoch_itr.load_state_dict(extra["train_iterator"])
        log(event="resume", file=restore, num_updates=trainer.num_updates, epoch=epoch_itr.epoch)
    else:
        extra = None

    max_update = args.max_update or float("inf")
    max_epoch = args.max_epoch or float("inf")
    summary = {"train_loss": [], "valid": [], "saved": []}
    # fairseq/checkpoint_utils.py:41-46, 262-264: the best validation score survives a resume (a worse first validation
    # after it must not overwrite checkpoint_best.pt)
    # (restored only when neither --reset-optimizer nor --reset-meters is given: checkpoint_utils.py:256-262)
    best = extra.get("best") if (extra and not args.reset_optimizer and not getattr(args, "reset_meters", False)) else None

    # position of the TRAINER inside the epoch (the background stager runs ahead of it inside epoch_itr's own iterator,
    # so the iterator's counter is not the resume point)
    pos = {"consumed": 0, "total": 0}

    def iter_state():
        if pos["total"] and pos["consumed"] >= pos["total"]:
            return {"version": 2, "epoch": epoch_itr.epoch + 1, "iterations_in_epoch": 0, "shuffle": True}
        return {"version": 2, "epoch": epoch_itr.epoch, "iterations_in_epoch": pos["consumed"], "shuffle": True}

    def prune_checkpoints():
        """fairseq/checkpoint_utils.py:137-187: --keep-last-epochs N keeps the N newest checkpoint<epoch>.pt,
        --keep-best-checkpoints N the N best checkpoint.best_<metric>_<value>.pt."""
        import re
        def by_pattern(pattern):
            rx, found = re.compile(pattern), []
            for fn in os.listdir(args.save_dir):
                m = rx.fullmatch(fn)
                if m:
                    found.append((float(m.group(1)), fn))
            return [fn for _, fn in sorted(found, reverse=True)]
        if args.keep_last_epochs > 0:
            for fn in by_pattern(r"checkpoint(\d+)\.pt")[args.keep_last_epochs:]:
                os.remove(os.path.join(args.save_dir, fn))
        if args.keep_best_checkpoints > 0:
            names = by_pattern(r"checkpoint\.best_%s_(\d+\.?\d*)\.pt" % re.escape(args.best_checkpoint_metric))
            if not args.maximize_best_checkpoint_metric:
                names = names[::-1]
            for fn in names[args.keep_best_checkpoints:]:
                os.remove(os.path.join(args.save_dir, fn))

    def save(tag_files: List[str], val: Optional[Dict[str, float]]):
        nonlocal best
        if args.no_save:
            return
        # every rank: a skipped (non-finite) update raises on all of them together -- rank 0 raising alone would leave
        # the others blocked in the next all-reduce
        trainer.check_overflow()
        if rank != 0:
            return
        state_extra = {"train_iterator": iter_state(), "val_loss": (val or {}).get(args.best_checkpoint_metric)}
        files = list(tag_files)
        if val and args.best_checkpoint_metric in val:
            v = val[args.best_checkpoint_metric]
            best, names = best_checkpoint_files(os.listdir(args.save_dir), args.best_checkpoint_metric, args.keep_best_checkpoints,
                                                args.maximize_best_checkpoint_metric, v, best, epoch_itr.epoch, trainer.num_updates)
            files += names
        state_extra["best"] = best
        for fn in files:
            path = os.path.join(args.save_dir, fn)
            checkpoint_utils.save_checkpoint(path, trainer, state_extra)
            summary["saved"].append(path)
        prune_checkpoints()

    def validate_and_save(end_of_epoch: bool):
        """fairseq_cli/train.py:322-396."""
        nu = trainer.num_updates
        do_save = (end_of_epoch and epoch_itr.epoch % args.save_interval == 0) or \
                  (args.save_interval_updates > 0 and nu > 0 and nu % args.save_interval_updates == 0)
        do_validate = bool(valid_subsets) and nu >= args.validate_after_updates and (
            (end_of_epoch and epoch_itr.epoch % args.validate_interval == 0) or do_save or
            (args.validate_interval_updates > 0 and nu > 0 and nu % args.validate_interval_updates == 0))
        val = None
        if do_validate:
            val = validate(args, trainer, task, valid_subsets, world, rank)
            summary["valid"].append(dict(val, num_updates=nu))
            log(event="valid", num_updates=nu, epoch=epoch_itr.epoch, **{k: round(float(v), 5) for k, v in val.items()})
        if do_save:
            names = ["checkpoint_last.pt"]
            if end_of_epoch:
                names.insert(0, f"checkpoint{epoch_itr.epoch}.pt")
            else:
                names.insert(0, f"checkpoint_{epoch_itr.epoch}_{nu}.pt")
            save(names, val)

    t0 = time.perf_counter()
    window: List = []
    epoch_idx = epoch_itr.next_epoch_idx
    while trainer.num_updates < max_update and epoch_idx <= max_epoch:
        itr = epoch_itr.next_epoch_itr(shuffle=True)
        pos["consumed"], pos["total"] = itr.n, len(itr)
        feed = DevicePrefetcher(itr, trainer.engine, depth=max(2, args.num_workers),
                                model=model if model.hubert is not None else None)
        group: List = []
        for sample in feed:
            group.append(sample)
            pos["consumed"] += 1
            if len(group) < args.update_freq and pos["consumed"] < pos["total"]:
                continue
            # forward, backward, gradient exchange, clip, Adam: no host sync.  The update overlaps the next step's forward
            # (S2ST_ADAM_OVERLAP=0 switches that off); every other reader of parameters / optimizer outputs below goes
            # through trainer.wait_optimizer().  Round 6: 0.02 - 0.11 ms per step with the nontemporal optimizer kernel
            r = trainer.train_step(group, overlap_optimizer=os.environ.get("S2ST_ADAM_OVERLAP", "1") != "0")
            group = []
            window.append(r)
            nu = trainer.num_updates
            if nu % args.log_interval == 0 or nu >= max_update:
                trainer.check_overflow()  # FloatingPointError like fairseq/trainer.py:860-867
                logs = [lg for x in window for lg in x["logs"]]
                red = criterion.__class__.reduce_metrics(logs) if logs else {}
                dt = time.perf_counter() - t0
                log(event="train", num_updates=nu, epoch=epoch_itr.epoch, lr=r["lr"], gnorm=round(float(r["gnorm"][0]), 5),
                    ups=round(len(window) / max(dt, 1e-9), 3), **{k: round(float(v), 5) for k, v in red.items()})
                summary["train_loss"].append((nu, float(red.get("loss", float("nan")))))
                window, t0 = [], time.perf_counter()
            end = pos["consumed"] >= pos["total"]
            validate_and_save(end_of_epoch=end)
            if nu >= max_update:
                break
        feed.close()
        epoch_idx = epoch_itr.epoch + 1 if pos["consumed"] >= pos["total"] else epoch_itr.epoch
        if trainer.num_updates >= max_update:
            break
    trainer.check_overflow()
    if not args.no_save and rank == 0 and not summary["saved"]:
        save(["checkpoint_last.pt"], None)
    summary["num_updates"] = trainer.num_updates
    summary["trainer"] = trainer
    log(event="done", num_updates=trainer.num_updates)
    return summary


def cli_main():
    main(sys.argv[1:])


if __name__ == "__main__":
    cli_main()
