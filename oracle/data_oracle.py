"""CPU restatement of the data path's integer work -- TEST INFRASTRUCTURE (only tests/ may import it).

``batch_by_size``: behavioural restatement of ``batch_by_size_vec`` (fairseq/data/data_utils_fast.pyx:20-100).
Pinned against the reference's own Cython extension built into oracle/_ref by oracle/build_ref.sh (when the
reference is present) and against tests/golden/batcher.npz generated from that build (oracle/gen_golden_data.py).
"""
from typing import List

import numpy as np


def batch_by_size(
    indices: np.ndarray,
    num_tokens_vec: np.ndarray,
    max_tokens: int,
    max_sentences: int = 0,
    bsz_mult: int = 1,
) -> List[np.ndarray]:
    """Greedy length-bucketed packing: cost(batch) = len(batch) * max(tokens).

    Behavioural restatement of ``batch_by_size_vec``
    (fairseq/data/data_utils_fast.pyx:20-100): a running batch plus a tail; the tail is
    merged into the batch whenever the merged size is < bsz_mult or a multiple of it; on
    overflow of max_tokens / max_sentences the batch is closed and the tail starts the next.
    """
    n = len(indices)
    if n == 0:
        return []
    assert max_tokens <= 0 or int(np.max(num_tokens_vec)) <= max_tokens
    ends = np.zeros(n + 1, dtype=np.int64)
    count = 0
    batch_start = 0
    tail_max = 0
    batch_max = 0
    for pos in range(n):
        tail_max = max(tail_max, int(num_tokens_vec[pos]))
        new_end = pos + 1
        new_max = max(batch_max, tail_max)
        new_sent = new_end - batch_start
        new_tok = new_sent * new_max
        overflow = (max_sentences > 0 and new_sent > max_sentences) or (
            max_tokens > 0 and new_tok > max_tokens
        )
        fits_mult = new_sent < bsz_mult or new_sent % bsz_mult == 0
        if overflow:
            tail_tok = tail_max * (new_end - ends[count])
            if max_tokens > 0 and tail_tok > max_tokens:
                count += 1
                ends[count] = pos
                tail_max = int(num_tokens_vec[pos])
            batch_start = int(ends[count])
            count += 1
            new_max = tail_max
        if overflow or fits_mult:
            ends[count] = new_end
            batch_max = new_max
            tail_max = 0
    if ends[count] != n:
        count += 1
    return [b for b in np.split(np.asarray(indices), ends[:count]) if len(b) > 0]


