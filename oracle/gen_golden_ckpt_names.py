#!/usr/bin/env python3
"""Golden of the checkpoint FILE NAMES the reference writes and keeps for a sequence of validation scores (build container
only):   python oracle/gen_golden_ckpt_names.py     # writes tests/golden/ckpt_names.npz
TEST INFRASTRUCTURE.  Drives fairseq/checkpoint_utils.py:34-187 (``save_checkpoint``) itself with a stub trainer whose
``save_checkpoint(path, extra)`` just creates the file: checkpoint_best / checkpoint.best_<metric>_<score><digit> /
checkpoint_last / per-epoch files, --keep-best-checkpoints pruning, for a minimised and a maximised metric."""
import os
import sys
import tempfile
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")
for _n, _t in dict(float=float, int=int, bool=bool, object=object, complex=complex, str=str).items():
    if not hasattr(np, _n):
        setattr(np, _n, _t)
torch._C.has_cudnn = False
import fairseq  # noqa: E402,F401
from fairseq import checkpoint_utils as CU  # noqa: E402

SCORES = [3.25, 2.9004, 2.9004, 3.0, 2.5, 2.75, 2.4996, 2.4996, 2.2, 4.0]


class Trainer:
    data_parallel_rank = 0
    should_save_checkpoint_on_current_rank = True
    always_call_state_dict_during_save_checkpoint = False
    checkpoint_suffix = ""

    def __init__(self):
        self.updates = 0

    def get_num_updates(self):
        return self.updates

    def consolidate_optimizer(self):
        pass

    def save_checkpoint(self, path, extra):
        open(path, "w").write("x")


class Itr:
    def __init__(self):
        self.epoch = 1

    def end_of_epoch(self):
        return True

    def state_dict(self):
        return {}


def run(maximize, keep):
    out = []
    with tempfile.TemporaryDirectory() as d:
        cfg = Namespace(save_dir=d, maximize_best_checkpoint_metric=maximize, no_save=False, no_epoch_checkpoints=True,
                        save_interval=1, save_interval_updates=0, keep_best_checkpoints=keep, best_checkpoint_metric="loss",
                        no_last_checkpoints=False, write_checkpoints_asynchronously=False, keep_interval_updates=-1,
                        keep_interval_updates_pattern=-1, keep_last_epochs=-1)
        if hasattr(CU.save_checkpoint, "best"):
            del CU.save_checkpoint.best
        tr, it = Trainer(), Itr()
        for k, v in enumerate(SCORES):
            tr.updates = 10 * (k + 1)
            it.epoch = k + 1
            before = {f: os.path.getmtime(os.path.join(d, f)) for f in os.listdir(d)}
            for f in before:  # so that a rewrite is visible
                os.utime(os.path.join(d, f), (1, 1))
            CU.save_checkpoint(cfg, tr, it, v)
            after = sorted(os.listdir(d))
            written = sorted(f for f in after if os.path.getmtime(os.path.join(d, f)) > 1)
            out.append((written, after, float(CU.save_checkpoint.best)))
    return out


def main():
    rec = {"scores": np.asarray(SCORES)}
    for maximize in (False, True):
        for keep in (2, 3):
            r = run(maximize, keep)
            tag = f"{'max' if maximize else 'min'}.keep{keep}"
            for k, (w, a, b) in enumerate(r):
                rec[f"{tag}.{k}.written"] = np.asarray(w)
                rec[f"{tag}.{k}.listing"] = np.asarray(a)
                rec[f"{tag}.{k}.best"] = np.asarray(b)
            print(tag, [x[0] for x in r][-3:], r[-1][1])
    dst = os.path.join(os.path.dirname(HERE), "tests", "golden", "ckpt_names.npz")
    np.savez_compressed(dst, **rec)
    print("wrote", dst, os.path.getsize(dst))


if __name__ == "__main__":
    main()
