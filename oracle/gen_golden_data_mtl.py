#!/usr/bin/env python3
"""Golden outputs of the ``s2s_translation_mtl`` task's on-disk data path, produced by the REFERENCE's separate dataset
module examples/s2s_trans/data/s2st_dataset_mtl.py (build container only).

    python oracle/gen_golden_data_mtl.py        # writes tests/golden/data_path_mtl.npz

TEST INFRASTRUCTURE.  Same miniature corpus as oracle/gen_golden_data.py (tests/data_corpus.py) + one manifest with the
duration / pitch / energy columns; runs the reference's ``S2STDatasetCreator.from_tsv``, ``dataset[i]``,
``ordered_indices``, ``size`` and ``collater``.  Only numbers and the corpus' own words are stored.
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
for _n, _t in dict(float=float, int=int, bool=bool, object=object, complex=complex, str=str).items():
    if not hasattr(np, _n):
        setattr(np, _n, _t)
torch._C.has_cudnn = False

import fairseq  # noqa: E402,F401
from fairseq.data import Dictionary  # noqa: E402
from examples.s2s_trans.data.data_cfg import S2STDataConfig  # noqa: E402
from examples.s2s_trans.data.s2st_dataset_mtl import S2STDatasetCreator  # noqa: E402

from data_corpus import flatten_batch, make_corpus, make_mtl_extras  # noqa: E402

SCRATCH = "/tmp/s2st_data_corpus"  # the same absolute path the tests use (paths are written into the manifests)


def main():
    root = make_mtl_extras(make_corpus(SCRATCH))
    cfg = S2STDataConfig(Path(root) / "config.yaml")
    sd = Dictionary.load(os.path.join(root, cfg.src_vocab_filename))
    td = Dictionary.load(os.path.join(root, cfg.tgt_vocab_filename))
    out = {}
    for split in ("train_tiny", "dev_tiny", "dev_fs"):
        ds = S2STDatasetCreator.from_tsv(root, cfg, split, sd, td, None, None, is_train_split=split.startswith("train"),
                                         epoch=1, seed=1, n_frames_per_step=4, speaker_to_id={"spk0": 0, "spk1": 1})
        np.random.seed(11)  # SpecAugment (train split) draws from numpy's global RNG
        items = [ds[i] for i in range(len(ds))]
        for i, it in enumerate(items):
            # (the feature side is the base dataset's code, pinned array by array in data_path.npz: a checksum here)
            out[f"{split}.item{i}.speech_sums"] = np.asarray([it.src_speech.double().sum().item(), it.tgt_speech.double().sum().item(),
                                                             it.src_speech.shape[0], it.tgt_speech.shape[0]])
            out[f"{split}.item{i}.src_text"] = it.src_text.numpy()
            out[f"{split}.item{i}.tgt_text"] = it.tgt_text.numpy()
            if it.duration is not None:
                out[f"{split}.item{i}.duration"] = it.duration.numpy()
                out[f"{split}.item{i}.pitch"] = it.pitch.numpy()
                out[f"{split}.item{i}.energy"] = it.energy.numpy()
        out[f"{split}.ordered_indices"] = np.asarray(ds.ordered_indices())
        out[f"{split}.sizes"] = np.asarray([ds.size(i) for i in range(len(ds))])
        pick = [4, 0, 7, 2] if len(ds) > 7 else [1, 3, 0]
        for k, v in flatten_batch(ds.collater([items[i] for i in pick])).items():
            out[f"{split}.batch.{k}"] = v
        out[f"{split}.batch_pick"] = np.asarray(pick)
    dst = os.path.join(ROOT, "tests", "golden", "data_path_mtl.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, len(out), "arrays,", os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
