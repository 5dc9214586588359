"""One pool of HIP streams per device, shared by every object of the package that runs work beside the caller's stream.

Why a pool (round 5).  ROCm maps a process's streams onto a small number of hardware queues (``GPU_MAX_HW_QUEUES``, 4 by
default) in creation order.  Two streams that share a hardware queue run one after the other -- and inherit each other's
waits.  Rounds 3 - 4 created a stream wherever one was needed (a vocoder's run-ahead stream, its copy stream, a generator's
chain streams, a twin engine's never-used second stream ...), so which streams collided depended on how many objects a
program had built before: ``bench.py --config infer_base`` measured 390 - 460 or 760 - 890 utterances/s for the SAME leg
depending on the legs run before it (profiles/r05_infer_chains_stream_aliasing.txt).  With one stream per ROLE, created once
and handed to whoever plays that role, the set of streams a workload uses -- hence its queue mapping -- no longer depends on
object counts: decode chains 1 .. n, the deferred vocoder, the phase-draw upload, the frozen front end, the gradient
exchange, the batch prefetcher.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

_POOL: Dict[Tuple[int, str], "torch.cuda.Stream"] = {}


def get(role: str, device=None) -> "torch.cuda.Stream":
    """The device's stream for ``role`` (created on first use, then kept for the life of the process)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, role)
    st = _POOL.get(key)
    if st is None:
        st = _POOL[key] = torch.cuda.Stream(device=torch.device("cuda", idx))
    return st
