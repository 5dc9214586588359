#!/usr/bin/env python3
"""Golden outputs of the on-disk data path, produced by the REFERENCE dataset classes (build container only).

    python oracle/gen_golden_data.py        # writes tests/golden/data_path.npz

TEST INFRASTRUCTURE.  Imports /root/reference (stub packages in oracle/ref_shims, numpy alias prelude), writes the
miniature corpus of tests/data_corpus.py into a scratch directory and runs the reference's ``S2STDataConfig``,
``Dictionary.load``, ``S2STDatasetCreator.from_tsv``, ``dataset[i]``, ``ordered_indices`` and ``collater`` on it.
Only numbers are stored; no reference source enters the repo.
"""
import os
import sys
import tempfile
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
from examples.s2s_trans.data.s2st_dataset import S2STDatasetCreator  # noqa: E402

from data_corpus import flatten_batch, make_corpus  # noqa: E402

SCRATCH = "/tmp/s2st_data_corpus"  # the same absolute path the test uses (paths are written into the manifests)


def batcher_cases():
    """(name, num_tokens sorted descending, max_tokens, max_sentences, bsz_mult) -- incl. the tail-overflow corner."""
    rs = np.random.RandomState(5)
    cases = []
    for k, (n, lo, hi, mt, ms, mult) in enumerate([(200, 40, 3000, 20000, -1, 8), (64, 10, 400, 1000, -1, 1),
                                                  (97, 1, 50, 120, 7, 4), (33, 100, 101, 1000, -1, 8),
                                                  (50, 5, 900, 900, 3, 2), (1, 7, 8, 100, -1, 8),
                                                  (40, 20, 500, 0, 6, 4), (120, 30, 2000, 2000, -1, 16)]):
        nt = np.sort(rs.randint(lo, hi, size=n))[::-1].astype(np.int64)
        if k == 2:
            nt = rs.randint(lo, hi, size=n).astype(np.int64)  # unsorted order exercises the tail-overflow branch
        cases.append((f"case{k}", nt, mt, ms, mult))
    return cases


def batcher_golden():
    """Outputs of the reference's own Cython batcher (built into oracle/_ref by oracle/build_ref.sh)."""
    import subprocess
    subprocess.check_call([os.path.join(HERE, "build_ref.sh")])
    sys.path.insert(0, os.path.join(HERE, "_ref"))
    import data_utils_fast as F
    out = {}
    for name, nt, mt, ms, mult in batcher_cases():
        idx = np.arange(len(nt), dtype=np.int64)
        b = F.batch_by_size_vec(idx, nt, mt, ms, mult)
        out[name + ".num_tokens"] = nt
        out[name + ".args"] = np.asarray([mt, ms, mult])
        out[name + ".ends"] = np.cumsum([len(x) for x in b]).astype(np.int64)
    dst = os.path.join(ROOT, "tests", "golden", "batcher.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


def iterator_golden():
    """Batch order per epoch / shard from the reference's EpochBatchIterator (+ resume position)."""
    from fairseq.data import iterators as ref_it
    sys.path.insert(0, os.path.join(HERE, "_ref"))
    import data_utils_fast as F
    name, nt, mt, ms, mult = batcher_cases()[0]
    batches = [b.tolist() for b in F.batch_by_size_vec(np.arange(len(nt), dtype=np.int64), nt, mt, ms, mult)]

    class DS(torch.utils.data.Dataset):
        def __getitem__(self, i):
            return int(i)

        def __len__(self):
            return len(nt)

    out = {"n_batches": np.asarray(len(batches))}
    for shards in (1, 3):
        for sid in range(shards):
            it = ref_it.EpochBatchIterator(DS(), lambda x: list(x), batches, seed=3, num_shards=shards, shard_id=sid, epoch=1)
            for ep in (1, 2, 3):
                seq = list(it.next_epoch_itr(shuffle=True))
                out[f"s{shards}.{sid}.e{ep}.flat"] = np.asarray([i for b in seq for i in b], dtype=np.int64)
                out[f"s{shards}.{sid}.e{ep}.lens"] = np.asarray([len(b) for b in seq], dtype=np.int64)
                assert it.end_of_epoch()
            st = it.state_dict()
            out[f"s{shards}.{sid}.final_state"] = np.asarray([st["epoch"], st["iterations_in_epoch"]])
    it = ref_it.EpochBatchIterator(DS(), lambda x: list(x), batches, seed=3, num_shards=3, shard_id=1, epoch=1)
    itr = it.next_epoch_itr(shuffle=True)
    for _ in range(2):
        next(itr)
    st = it.state_dict()
    out["resume.state"] = np.asarray([st["epoch"], st["iterations_in_epoch"]])
    it2 = ref_it.EpochBatchIterator(DS(), lambda x: list(x), batches, seed=3, num_shards=3, shard_id=1, epoch=1)
    it2.load_state_dict(st)
    rest = list(it2.next_epoch_itr(shuffle=True))
    out["resume.flat"] = np.asarray([i for b in rest for i in b], dtype=np.int64)
    out["resume.lens"] = np.asarray([len(b) for b in rest], dtype=np.int64)
    dst = os.path.join(ROOT, "tests", "golden", "epoch_iterator.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


def main():
    root = make_corpus(SCRATCH)
    cfg = S2STDataConfig(Path(root) / "config.yaml")
    cfg.set_use_hubert(False)
    cfg.set_kd_encoder(False)
    sd = Dictionary.load(os.path.join(root, cfg.src_vocab_filename))
    td = Dictionary.load(os.path.join(root, cfg.tgt_vocab_filename))
    out = {"src_dict_len": np.asarray(len(sd)), "tgt_dict_len": np.asarray(len(td))}
    for split in ("train_tiny", "dev_tiny"):
        ds = S2STDatasetCreator.from_tsv(root, cfg, split, sd, td, None, None, is_train_split=split.startswith("train"),
                                         epoch=1, seed=1, n_frames_per_step=4, speaker_to_id={"spk0": 0, "spk1": 1})
        np.random.seed(11)  # SpecAugment (train split) draws from numpy's global RNG
        items = [ds[i] for i in range(len(ds))]
        for i, it in enumerate(items):
            out[f"{split}.item{i}.src_speech"] = it.src_speech.numpy()
            out[f"{split}.item{i}.tgt_speech"] = it.tgt_speech.numpy()
            out[f"{split}.item{i}.src_text"] = it.src_text.numpy()
            out[f"{split}.item{i}.tgt_text"] = it.tgt_text.numpy()
        out[f"{split}.ordered_indices"] = np.asarray(ds.ordered_indices())
        out[f"{split}.sizes"] = np.asarray([ds.size(i) for i in range(len(ds))])
        pick = [4, 0, 7, 2] if len(ds) > 7 else [1, 3, 0]
        for k, v in flatten_batch(ds.collater([items[i] for i in pick])).items():
            out[f"{split}.batch.{k}"] = v
        out[f"{split}.batch_pick"] = np.asarray(pick)
    batcher_golden()
    iterator_golden()
    dst = os.path.join(ROOT, "tests", "golden", "data_path.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, len(out), "arrays,", os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
