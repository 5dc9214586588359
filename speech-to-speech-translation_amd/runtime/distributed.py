"""Data-parallel gradient exchange: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI; "gloo" in the CPU tests).

Replaces torch DDP as constructed at fairseq/models/distributed_fairseq_model.py:58-67 (bucketed
all-reduce overlapped with backward) and the statistics exchange of
fairseq/trainer.py:1297-1323, 1365-1372 (pickled logging outputs, grad-norm vector): here the
parameter arena is laid out in forward-use order, so the backward finalises contiguous
ranges back-to-front; each finished range is all-reduced on a side stream while the
remaining backward runs.  xGMI is point-to-point (7 links/GPU), so ranges are kept large
(>= ~16 MiB) rather than DDP's 25 MB buckets tuned for NVSwitch rings.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


class NativeComm:
    """RCCL communicator behind the C ABI (``s2st_comm_*`` / ``s2st_allreduce_sum_f32``, include/s2st_hip.h): the
    128-byte unique id travels over the already-initialised ``torch.distributed`` group (any backend), the collective
    itself is issued by libs2st_hip.so on the caller's HIP stream."""

    def __init__(self):
        import ctypes as C
        from . import binding as bd
        self.C, self.bd, self.lib = C, bd, bd.lib()
        self.lib.s2st_comm_unique_id.argtypes = [C.c_void_p]
        self.lib.s2st_comm_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        self.lib.s2st_allreduce_sum_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        self.lib.s2st_comm_destroy.argtypes = [C.c_void_p]
        if not self.lib.s2st_comm_available():
            raise bd.S2STHipError("no RCCL library could be bound (s2st_comm_available() == 0)")
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        buf = C.create_string_buffer(128)
        if rank == 0:
            bd.check(self.lib.s2st_comm_unique_id(buf), "s2st_comm_unique_id")
        if world > 1:
            box = [bytes(buf.raw)]
            dist.broadcast_object_list(box, src=0)
            buf = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        bd.check(self.lib.s2st_comm_init(buf, world, rank, C.byref(h)), "s2st_comm_init")
        self.h, self.world, self.rank = h, world, rank

    def all_reduce_(self, t: torch.Tensor):
        """In-place SUM over ranks, ordered on the current torch stream."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self.bd.check(self.lib.s2st_allreduce_sum_f32(self.h, t.data_ptr(), t.numel(), self.bd.stream_ptr()),
                      "s2st_allreduce_sum_f32")

    def close(self):
        if getattr(self, "h", None):
            self.lib.s2st_comm_destroy(self.h)
            self.h = None


class GradReducer:
    """SUM all-reduce of arena ranges as the backward finishes them.

    Three transports, same ordering logic (a range is reduced on the reducer's stream behind BOTH engine streams):
    * device gradients, backend "nccl" (= RCCL): ``dist.all_reduce`` on the reducer's stream, or -- with
      ``S2ST_NATIVE_ALLREDUCE=1`` -- the C ABI's own communicator (``NativeComm``);
    * device gradients, backend "gloo": the range is staged through pinned host memory around a host all-reduce.  Slow
      (it blocks the host), but it lets two ranks share ONE GPU, which RCCL refuses ("Duplicate GPU detected"): the
      two-process GPU test of the exchange runs this way;
    * host gradients (the CPU emulator tests): asynchronous gloo all-reduce."""

    def __init__(self, grads: torch.Tensor, min_bucket_floats: int = 4 << 20, extra_stream=None, exchange_dtype: str = "fp32"):
        """``exchange_dtype``: "fp32" (default: what the reference's DDP sums, distributed_fairseq_model.py:58-67) or "bf16"
        (``--grad-exchange-dtype bf16``): a finished range is rounded to bf16 by ``s2st_grad_pack_bf16_f32``, summed over
        the ranks in that type, and widened back into the fp32 arena -- half the wire bytes, at the price of
        one rounding of every rank's contribution and bf16 partial sums inside the collective; the two-rank trajectory test
        bounds the effect (tests/test_distributed.py)."""
        self.grads = grads
        if exchange_dtype not in ("fp32", "bf16"):
            raise ValueError("--grad-exchange-dtype must be fp32 or bf16")
        self.exchange_dtype = exchange_dtype
        self._half = None
        import os
        self.staged = grads.is_cuda and dist.is_initialized() and dist.get_backend() == "gloo"
        self.native = None
        if grads.is_cuda and not self.staged and os.environ.get("S2ST_NATIVE_ALLREDUCE", "0") == "1":
            self.native = NativeComm()
        self._pin = None
        # the engine's second stream (weight gradients): a range is final once both streams reached here
        self.extra_stream = extra_stream
        self.min_bucket = min_bucket_floats
        self.cuda = grads.is_cuda
        from . import streams
        # (its own hardware queue: not the engine's weight-gradient stream's, whose ranges it ships while that stream works on
        #  the next ones; the batch prefetcher's rare copies may sit behind it)
        self.stream = (streams.get("gradient-exchange", grads.device, avoid=[extra_stream] if extra_stream is not None else None,
                                   may_share=("prefetch",)) if self.cuda else None)
        self.pending: Optional[Tuple[int, int]] = None
        self.handles: List = []
        self.reduced: List[Tuple[int, int]] = []
        # instrumentation (bench.py --gpus N): how long the compute stream waits in finish() for the exchange -- the part
        # of the all-reduce that the backward did NOT hide -- and the ranges of the last update
        self.measure_exposed = False
        self._exposed_events: List = []
        self.last_buckets: List[Tuple[int, int]] = []
        # S2ST_EXCHANGE_PROXY="<workgroups>,<ranks>,<GB/s>" on ONE GPU (bench.py --exchange-proxy): no collective, but every
        # bucket launches the library's stand-in for the collective's kernels where the all-reduce would go -- that many
        # workgroups reading + writing the 2 (N - 1) / N share of the bucket at the given pace (s2st_exchange_proxy_f32) --
        # so that a single GPU can price what RCCL's kernels beside the backward cost the step
        self.proxy = None
        spec = os.environ.get("S2ST_EXCHANGE_PROXY")
        if spec and self.cuda and not dist.is_initialized():
            w, n, g = spec.split(",")
            self.proxy = (int(w), int(n), float(g))
            self._proxy_scratch = None

    def _launch(self, lo: int, hi: int):
        if hi <= lo:
            return
        view = self.grads[lo:hi]
        self.reduced.append((lo, hi))
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            if self.extra_stream is not None:
                self.stream.wait_stream(self.extra_stream)
            with torch.cuda.stream(self.stream):
                if self.proxy is not None:
                    from . import binding as bd
                    wgs, nranks, gbps = self.proxy
                    if self._proxy_scratch is None or self._proxy_scratch.numel() < view.numel():
                        self._proxy_scratch = torch.empty(max(view.numel(), self.min_bucket), dtype=torch.float32, device=view.device)
                    esz = 2 if self.exchange_dtype == "bf16" else 4
                    move = int(2 * (nranks - 1) / nranks * view.numel() * esz)
                    bd.call("s2st_exchange_proxy_f32", view, self._proxy_scratch, view.numel(), move, wgs, gbps)
                elif self.exchange_dtype == "bf16":
                    self._exchange_bf16(view)
                elif self.staged:
                    if self._pin is None or self._pin.numel() < view.numel():
                        self._pin = torch.empty(max(view.numel(), self.min_bucket), dtype=torch.float32).pin_memory()
                    host = self._pin[:view.numel()]
                    host.copy_(view, non_blocking=True)
                    self.stream.synchronize()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM)
                    view.copy_(host, non_blocking=True)
                elif self.native is not None:
                    self.native.all_reduce_(view)
                else:
                    dist.all_reduce(view, op=dist.ReduceOp.SUM)
        elif self.exchange_dtype == "bf16":
            self._exchange_bf16(view)
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def _exchange_bf16(self, view: torch.Tensor):
        """One range through the bf16 exchange, on the current stream (the reducer's own on a GPU; ranges follow each other
        on it, so one staging buffer serves them all)."""
        from . import binding as bd
        n = view.numel()
        if self._half is None or self._half.numel() < n:
            self._half = torch.empty(max(n, self.min_bucket), dtype=torch.bfloat16, device=view.device)
        half = self._half[:n]
        bd.call("s2st_grad_pack_bf16_f32", view, half, n)
        if self.staged:  # two ranks on one GPU (tests): gloo carries the bytes through pinned host memory
            if self._pin is None or self._pin.numel() * 2 < n:
                self._pin = torch.empty(max(n, self.min_bucket), dtype=torch.float32).pin_memory()
            host = self._pin.view(torch.bfloat16)[:n]
            host.copy_(half, non_blocking=True)
            self.stream.synchronize()
            self._host_sum_bf16(host)
            half.copy_(host, non_blocking=True)
        elif view.is_cuda:
            dist.all_reduce(half, op=dist.ReduceOp.SUM)
        else:
            self._host_sum_bf16(half)
        bd.call("s2st_grad_unpack_bf16_f32", half, view, n)

    @staticmethod
    def _host_sum_bf16(t: torch.Tensor):
        """gloo has no bf16 reduction on every build: gather the ranks' bf16 buffers and add them in rank order with a
        bf16 running sum (what a ring reduction's hops hold)."""
        parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t)
        acc = parts[0].clone()
        for q in parts[1:]:
            acc += q
        t.copy_(acc)

    def on_segment(self, i: int, lo: int, hi: int):
        """Engine callback: gradients in [lo, hi) are final.  Ranges arrive back-to-front and
        are coalesced until they reach ``min_bucket``."""
        if not is_dist() and self.proxy is None:
            return
        if self.pending is None:
            self.pending = (lo, hi)
        else:
            plo, phi = self.pending
            assert hi == plo, "segments must arrive contiguously, back to front"
            self.pending = (lo, phi)
        if self.pending[1] - self.pending[0] >= self.min_bucket:
            self._launch(*self.pending)
            self.pending = None

    def finish(self):
        """Flush the tail bucket and make the reduced gradients visible to the compute stream."""
        if not is_dist() and self.proxy is None:
            return
        if self.pending is not None:
            self._launch(*self.pending)
            self.pending = None
        if self.cuda:
            cur = torch.cuda.current_stream()
            if self.measure_exposed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(cur)
                cur.wait_stream(self.stream)
                b.record(cur)
                self._exposed_events.append((a, b))
            else:
                cur.wait_stream(self.stream)
        for h in self.handles:
            h.wait()
        self.handles.clear()
        covered = sorted(self.reduced)
        self.last_buckets = list(self.reduced)  # launch order: back to front
        self.reduced = []
        return covered

    def exposed_ms(self) -> List[float]:
        """Per update since the last call: milliseconds the compute stream spent waiting for the exchange in finish()
        (synchronises the device)."""
        if not self._exposed_events:
            return []
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self._exposed_events]
        self._exposed_events = []
        return out


def all_reduce_scalars(t: torch.Tensor) -> torch.Tensor:
    """Fixed-layout fp32/fp64 statistics vector (sample sizes, loss sums): one small SUM
    all-reduce instead of the reference's pickled all_gather_list."""
    if is_dist():
        if t.is_cuda and dist.get_backend() == "gloo":  # two ranks on one GPU (tests): through the host
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Parameter / buffer broadcast at start-up (DDP's constructor does the same)."""
    if is_dist():
        if t.is_cuda and dist.get_backend() == "gloo":
            h = t.cpu()
            dist.broadcast(h, src)
            t.copy_(h)
        else:
            dist.broadcast(t, src)
    return t
