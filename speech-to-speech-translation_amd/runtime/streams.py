"""One pool of HIP streams per device, shared by every object of the package that runs work beside the caller's stream.

Why a pool (round 5).  ROCm maps a process's streams onto a small number of hardware queues (``GPU_MAX_HW_QUEUES``, 4 by
default; the caller's stream holds one of them).  Two streams that share a hardware queue run one after the other -- and
inherit each other's waits.  Rounds 3 - 4 created a stream wherever one was needed (a vocoder's run-ahead stream, its copy
stream, a generator's chain streams, a twin engine's never-used second stream ...), so which streams collided depended on
how many objects a program had built before: ``bench.py --config infer_base`` measured 390 - 460 or 760 - 890 utterances/s
for the SAME leg depending on the legs run before it (profiles/r05_infer_chains_stream_aliasing.txt).  With one stream per
ROLE, created once and handed to whoever plays that role, the set of streams a workload uses no longer depends on object
counts: decode chains 1 .. n, the deferred vocoder, the phase-draw upload, the frozen front end, the gradient exchange, the
batch prefetcher.

Which queue a new stream lands on is the runtime's business (tools/r05_queue_matrix.py, profiles/r05_queue_matrix.txt:
eleven streams fall into FOUR groups that run one after the other, with ``GPU_MAX_HW_QUEUES=8`` as well -- four queues is
what a process gets here -- and which stream joins the caller's group depends on that variable and on what was created
before), so the pool MEASURES it: a new role's stream is accepted only if a short chain of spin kernels on it runs beside
the same chain on every stream it must not share a queue with -- the caller's, every pool stream except the roles named
in ``may_share``, the ``avoid`` list -- instead of behind it (two streams on one queue take the sum of their chains, two
queues the maximum).  Up to ``_TRIES`` candidates are made per role; if none is free (more exclusive roles than queues)
the last one is kept and the collision is recorded (``collisions()``: bench.py reports it).  With four queues a workload
has to say which roles may double up: config 5 gives the caller's stream and two more decode chains a queue each and lets
the deferred vocoder and the phase-draw upload share the fourth.  ``S2ST_STREAM_PROBE=0`` switches the measurement off
(creation order decides, as before).
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional, Tuple

import torch

_POOL: Dict[Tuple[int, str], "torch.cuda.Stream"] = {}
_COLLISIONS: Dict[Tuple[int, str], str] = {}
_REJECTED: List["torch.cuda.Stream"] = []  # (kept alive: a destroyed stream's queue slot would be handed out again)
_TRIES = 10
_CHAIN, _SPIN = 24, 50000  # kernels per probe chain, cycles per kernel (~20 us: the host enqueues one in ~5)


def _chain_ms(streams) -> float:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            for _ in range(_CHAIN):
                torch.cuda._sleep(_SPIN)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def shares_queue(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
    """Do the two streams run one after the other?  (Both are used once first: a stream's first launch creates its
    hardware queue, milliseconds.)"""
    _chain_ms([a])
    _chain_ms([b])
    one = min(_chain_ms([a]), _chain_ms([b]))
    both = min(_chain_ms([a, b]), _chain_ms([a, b]))
    return both > 1.6 * one


# Roles of the TRAINING loop: a queue collision there costs speed, never results, and their streams are first asked for
# inside a training step (the gradient reducer of every DDP rank, the prefetcher's thread) -- a wall-clock probe with a
# device-wide synchronise does not belong there: creation order decides unless warm() is called for them.
_UNPROBED_ROLES = ("gradient-exchange", "prefetch", "front-end")


def _probe_on(role: str = "", explicit: bool = False) -> bool:
    if os.environ.get("S2ST_STREAM_PROBE", "1") == "0" or not hasattr(torch.cuda, "_sleep"):
        return False
    if not explicit and role in _UNPROBED_ROLES:
        return False
    # a probe synchronises the device: illegal while a stream of this thread is capturing a graph
    try:
        if torch.cuda.is_current_stream_capturing():
            return False
    except Exception:
        pass
    return True


def warm(roles, device=None, may_share=None) -> Dict[str, "torch.cuda.Stream"]:
    """Create (and probe) the streams of ``roles`` NOW -- at a defined point of a program's start-up (a generator's or
    trainer's construction), outside any timed region or graph capture -- instead of at their first use.  ``may_share``:
    {role: tuple of roles whose queue it may share}.  Results are cached per device like those of ``get``."""
    may_share = may_share or {}
    return {r: get(r, device, may_share=may_share.get(r, ()), _explicit=True) for r in roles}


def get(role: str, device=None, avoid: Optional[list] = None, may_share=(), _explicit: bool = False) -> "torch.cuda.Stream":
    """The device's stream for ``role`` (created on first use, then kept for the life of the process).  ``avoid``: further
    streams the role must not share a queue with (an engine's own second stream, say); ``may_share``: pool roles whose
    queue it may share (work that is rare or that runs after the other role's anyway)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, role)
    st = _POOL.get(key)
    if st is not None:
        return st
    d = torch.device("cuda", idx)
    if not _probe_on(role, _explicit):
        st = _POOL[key] = torch.cuda.Stream(device=d)
        return st
    with torch.cuda.device(d):
        others = [("the caller's stream", torch.cuda.current_stream(d))]
        others += [(r, s) for (i, r), s in _POOL.items() if i == idx and r not in may_share]
        others += [("avoid[%d]" % n, s) for n, s in enumerate(avoid or [])]
        hit = None
        for n_try in range(_TRIES):
            st = torch.cuda.Stream(device=d)
            hit = next((name for name, s in others if shares_queue(st, s)), None)
            if os.environ.get("S2ST_STREAM_PROBE_VERBOSE"):
                print(f"[streams] role {role}: candidate {n_try} (handle {st.cuda_stream:#x}) "
                      f"{'is free' if hit is None else 'shares a queue with ' + hit}; tested against {[n for n, _ in others]}",
                      flush=True)
            if hit is None:
                break
            if len(_REJECTED) < 2 * _TRIES:  # (bounded: beyond that a rejected candidate is simply dropped)
                _REJECTED.append(st)
        if hit is not None:
            _COLLISIONS[key] = hit
    _POOL[key] = st
    return st


def collisions(device=None) -> Dict[str, str]:
    """role -> the stream it shares a hardware queue with (roles for which no free queue was found)."""
    idx = None if device is None else torch.device(device).index
    return {r: other for (i, r), other in _COLLISIONS.items() if idx is None or i == idx}


def default_decode_chains() -> int:
    """How many batches ``generate_many`` callers decode at once by default: ``S2ST_DECODE_CHAINS``, else 3 -- the
    caller's stream and two chain streams on a queue each, vocoder + upload on the fourth."""
    if "S2ST_DECODE_CHAINS" in os.environ:
        return max(1, int(os.environ["S2ST_DECODE_CHAINS"]))
    return 3


def capture_stream(device=None) -> "torch.cuda.Stream":
    """The stream HIP graphs are captured on (never executes anything itself: not probed, takes no queue)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, "graph-capture")
    st = _POOL.get(key)
    if st is None:
        st = _POOL[key] = torch.cuda.Stream(device=torch.device("cuda", idx))
    return st
