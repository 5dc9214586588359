"""Upload the next batches while the GPU works on the current one.

``Engine.prepare`` turns a collated host batch into device-resident inputs (features, length / position vectors).
Its uploads are pageable copies: issued on the compute stream they synchronise with everything already enqueued, and
the host can no longer run ahead of the GPU (bench.py, ``S2ST_BENCH_VERBOSE=1``: 12.3 instead of 10.9 ms per step).
``DevicePrefetcher`` runs ``prepare`` for up to ``depth`` upcoming batches on a background thread and its own HIP
stream -- what fairseq's ``BufferedIterator`` + pinned ``DataLoader`` do for the reference
(fairseq/data/iterators.py:210-225, 565-640) -- and hands the trainer ``PreparedBatch`` objects.
"""
from __future__ import annotations

import queue
import threading
from typing import Dict, Iterable, Iterator, Optional

import torch


class PreparedBatch(tuple):
    """``(Batch, keep)`` pair of ``Engine.prepare`` that also answers the few dict look-ups the trainer makes on a
    sample (``ntokens``, ``nsentences``, ...)."""

    def __new__(cls, prepared, sample: Dict, ready: Optional["torch.cuda.Event"] = None):
        o = super().__new__(cls, prepared)
        o.sample, o.ready = sample, ready
        return o

    def __getitem__(self, k):
        if isinstance(k, str):
            return self.sample[k]
        return tuple.__getitem__(self, k)

    def __len__(self):  # a prepared batch is never the empty padding batch
        return 2

    def wait(self):
        """Make the current stream wait for the upload and keep the buffers alive for it."""
        if self.ready is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self.ready)
            for t in list(tuple.__getitem__(self, 1).values()) + list(getattr(self, "hubert_io", None) or ()):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)
            self.ready = None
        return self


class DevicePrefetcher:
    def __init__(self, batches: Iterable[Dict], engine, depth: int = 2, training: bool = True, model=None):
        """``model``: pass the S2STTransformerModel of a --use-hubert run so that batches are staged through
        ``model.prepare_sample`` (waveform upload + feature buffer) instead of ``engine.prepare``."""
        self.engine, self.training, self.model = engine, training, model
        self.q: "queue.Queue" = queue.Queue(maxsize=max(depth, 1))
        self.cuda = engine.device.type == "cuda"
        from . import streams
        self.stream = streams.get("prefetch", engine.device, may_share=("gradient-exchange",)) if self.cuda else None
        self._stop = False
        self._held = None  # (the look-ahead item of a --use-hubert run)
        self.thread = threading.Thread(target=self._run, args=(iter(batches),), daemon=True)
        self.thread.start()

    def _prepare(self, sample):
        if self.model is not None:
            return self.model.prepare_sample(sample, training=self.training)
        return PreparedBatch(self.engine.prepare(sample, training=self.training, seed=0), sample)  # (seed: set per step by forward)

    def _run(self, it: Iterator[Dict]):
        try:
            if self.cuda:
                torch.cuda.set_device(self.engine.device)
            for sample in it:
                if self._stop:
                    return
                if sample is None or len(sample) == 0:
                    self.q.put(sample)  # padding batch of a short shard: the trainer substitutes its dummy batch
                    continue
                if self.cuda:
                    with torch.cuda.stream(self.stream):
                        pb = self._prepare(sample)
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                        pb.ready = ev
                    self.q.put(pb)
                else:
                    self.q.put(self._prepare(sample))
            self.q.put(StopIteration)
        except BaseException as e:  # surfaced in the consumer
            self.q.put(e)

    def __iter__(self):
        return self

    def __next__(self):
        # --use-hubert: one batch of look-ahead -- when batch k is handed out, the frozen front end of batch k + 1 is
        # launched on the model's second stream (``front_end_ahead``), to run beside training step k
        ahead = self.model is not None and getattr(self.model, "hubert", None) is not None and self.cuda
        if self._held is not None:
            item, self._held = self._held, None
        else:
            item = self.q.get()
        if item is StopIteration:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        if ahead and isinstance(item, PreparedBatch):
            # (only when batch k + 1 is ALREADY staged: a step that is ready must not wait for a slow stager -- first
            # batches, validation, epoch tails; the front end of a batch that was not looked ahead runs in its own step)
            try:
                nxt = self.q.get_nowait()
            except queue.Empty:
                nxt = None
            if nxt is not None:
                self._held = nxt
                if isinstance(nxt, PreparedBatch):
                    self.model.front_end_ahead(nxt, after=nxt.ready)
        return item.wait() if isinstance(item, PreparedBatch) else item

    def close(self):
        self._stop = True
        while self.thread.is_alive():
            try:
                self.q.get_nowait()
            except queue.Empty:
                pass
            self.thread.join(timeout=0.05)
