"""Epoch iterator over max-tokens batches: per-epoch batch shuffling, sharding across data-parallel ranks,
resumable position.  Counterpart of ``fairseq/data/iterators.py:230-516`` (``EpochBatchIterator``), ``:518-548``
(``ShardedIterator``: rank r takes batches r, r + W, ...; short ranks are padded with EMPTY batches, for which the
trainer runs a dummy batch) and ``fairseq/data/data_utils.py:126-139`` (``numpy_seed``) -- every rank derives the
same shuffled order from ``seed + epoch`` with numpy's global RNG (state restored afterwards), so the ranks' batches
are exactly the reference's.  Single process, no worker pool: items are read when the batch is requested.
"""
from __future__ import annotations

import contextlib
import math
from typing import Callable, List, Optional, Sequence

import numpy as np


@contextlib.contextmanager
def numpy_seed(seed, *addl_seeds):
    if seed is None:
        yield
        return
    if len(addl_seeds) > 0:
        seed = int(hash((seed, *addl_seeds)) % 1e6)
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        yield
    finally:
        np.random.set_state(state)


def shard(batches: Sequence, num_shards: int, shard_id: int, fill_value=()):
    if shard_id < 0 or shard_id >= num_shards:
        raise ValueError("shard_id must be between 0 and num_shards")
    n = int(math.ceil(len(batches) / float(num_shards)))
    mine = list(batches[shard_id::num_shards])
    return mine + [fill_value] * (n - len(mine))


_WORKER_DATASET = None


def _loader_init(dataset_bytes):
    """Loader process start-up: its own copy of the dataset object (manifest columns, dictionaries, transform
    settings -- no features), one intra-op thread (the tensors handled here are KB-sized)."""
    global _WORKER_DATASET
    import pickle
    import torch
    torch.set_num_threads(1)
    _WORKER_DATASET = pickle.loads(dataset_bytes)


def _loader_collate(indices):
    ds = _WORKER_DATASET
    if len(indices) == 0:
        return {}
    return ds.collater([ds[int(i)] for i in indices])


class LoaderPool:
    """``--num-workers N`` loader PROCESSES (the reference's ``DataLoader(num_workers=4)``, fairseq/data/iterators.py:
    230-516): item loading is mostly interpreter work (manifest look-ups, .npy headers, dictionary encoding), which
    threads cannot overlap.  Spawned (not forked: the parent may hold an initialised HIP runtime), created once per
    iterator and reused across epochs; batches come back pickled, in submission order."""

    def __init__(self, dataset, num_workers: int):
        import pickle
        import torch.multiprocessing as mp  # tensors of a collated batch travel as shared-memory handles, not bytes
        self.pool = mp.get_context("spawn").Pool(num_workers, initializer=_loader_init, initargs=(pickle.dumps(dataset),))
        self.num_workers = num_workers
        self._pool = None

    def submit(self, indices):
        return self.pool.apply_async(_loader_collate, (list(map(int, indices)),))

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _CountingIterator:
    """In-order batches of one epoch; with a ``LoaderPool`` up to ``2 * num_workers`` batches are in flight ahead of
    the consumer and handed out in order."""

    def __init__(self, batches: List, collate: Callable, start: int = 0, pool: Optional["LoaderPool"] = None):
        self.batches, self.collate, self.n = batches, collate, start
        self.pool, self.pending = pool, {}
        if pool is not None:
            self.ahead = 2 * pool.num_workers
            self._submitted = start

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return self

    def has_next(self) -> bool:
        return self.n < len(self.batches)

    def __next__(self):
        if not self.has_next():
            raise StopIteration
        if self.pool is not None:
            while self._submitted < len(self.batches) and self._submitted < self.n + self.ahead:
                self.pending[self._submitted] = self.pool.submit(self.batches[self._submitted])
                self._submitted += 1
            out = self.pending.pop(self.n).get()
            self.n += 1
            return out
        b = self.batches[self.n]
        self.n += 1
        return self.collate(b)


class EpochBatchIterator:
    def __init__(self, dataset, collate_fn: Optional[Callable], batch_sampler: Sequence, seed: int = 1,
                 num_shards: int = 1, shard_id: int = 0, epoch: int = 1, disable_shuffling: bool = False,
                 num_workers: int = 0):
        self.num_workers = num_workers
        self._pool = None
        self.dataset = dataset
        self.collate_fn = collate_fn or (lambda items: dataset.collater(items))
        self.frozen_batches = tuple(batch_sampler)
        self.seed, self.num_shards, self.shard_id = seed, num_shards, shard_id
        self.epoch = max(epoch, 1)  # "we use 1-based indexing for epochs"
        self.disable_shuffling = disable_shuffling
        self.shuffle = not disable_shuffling
        self._cur_epoch_itr = None
        self._next_epoch_itr = None

    def __len__(self):
        return int(math.ceil(len(self.frozen_batches) / float(self.num_shards)))

    def _collate(self, indices):
        if len(indices) == 0:
            return {}  # padding batch of a short shard
        return self.collate_fn([self.dataset[int(i)] for i in indices])

    def epoch_batches(self, epoch: int, shuffle: bool) -> List:
        """The index batches this shard sees in ``epoch`` (what ``_get_iterator_for_epoch`` builds)."""
        batches = list(self.frozen_batches)
        if shuffle:
            with numpy_seed(self.seed + epoch):
                np.random.shuffle(batches)
        return shard(batches, self.num_shards, self.shard_id, fill_value=[])

    def _get_iterator_for_epoch(self, epoch, shuffle, offset=0):
        batches = self.epoch_batches(epoch, shuffle)
        if offset > 0 and offset >= len(batches):
            return None
        if self.num_workers > 0 and self._pool is None and hasattr(self.dataset, "collater"):
            self._pool = LoaderPool(self.dataset, self.num_workers)
        return _CountingIterator(batches, self._collate, start=offset, pool=self._pool)

    @property
    def next_epoch_idx(self):
        if self._next_epoch_itr is not None:
            return self.epoch
        if self._cur_epoch_itr is not None and self.end_of_epoch():
            return self.epoch + 1
        return self.epoch

    def next_epoch_itr(self, shuffle: bool = True):
        if self.disable_shuffling:
            shuffle = False
        self.epoch = self.next_epoch_idx
        if hasattr(self.dataset, "set_epoch"):
            self.dataset.set_epoch(self.epoch)
        if self._next_epoch_itr is not None:
            self._cur_epoch_itr, self._next_epoch_itr = self._next_epoch_itr, None
        else:
            self._cur_epoch_itr = self._get_iterator_for_epoch(self.epoch, shuffle)
        self.shuffle = shuffle
        return self._cur_epoch_itr

    def end_of_epoch(self) -> bool:
        return not self._cur_epoch_itr.has_next()

    @property
    def iterations_in_epoch(self):
        if self._cur_epoch_itr is not None:
            return self._cur_epoch_itr.n
        if self._next_epoch_itr is not None:
            return self._next_epoch_itr.n
        return 0

    def state_dict(self):
        if self._cur_epoch_itr is not None and self.end_of_epoch():
            epoch, it = self.epoch + 1, 0
        else:
            epoch, it = self.epoch, self.iterations_in_epoch
        return {"version": 2, "epoch": epoch, "iterations_in_epoch": it, "shuffle": self.shuffle}

    def load_state_dict(self, state_dict):
        self.epoch = state_dict["epoch"]
        pos = state_dict.get("iterations_in_epoch", 0)
        if pos > 0:
            self._next_epoch_itr = self._get_iterator_for_epoch(self.epoch, state_dict.get("shuffle", True), offset=pos)
            if self._next_epoch_itr is None:
                if state_dict.get("version", 1) == 1:
                    self.epoch += 1  # legacy: the epoch was finished
                else:
                    raise RuntimeError("Cannot resume training due to dataloader mismatch; relaunch with a reset "
                                       "data loader position")
        else:
            self._next_epoch_itr = None
