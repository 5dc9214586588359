"""Scorers the generation scripts report (``fairseq/scoring``): word error rate.

``WerScorer`` mirrors fairseq/scoring/wer.py:28-61 with its default configuration (tokenizer "none", no lowercasing, no
punctuation removal, word level): both strings are split on whitespace, the edit distance of the two token lists is
accumulated, ``score() = 100 * distance / reference length``.  The reference delegates the distance to the third-party
``editdistance`` package (un-pinned, absent from this image): ``editdistance.eval`` is the Levenshtein distance --
unit-cost insertions, deletions and substitutions -- restated here as the textbook two-row dynamic programme.
"""
from __future__ import annotations

from typing import Sequence


def edit_distance(a: Sequence, b: Sequence) -> int:
    """Levenshtein distance of two sequences (what ``editdistance.eval(a, b)`` returns)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, y in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y))
        prev = cur
    return prev[len(b)]


class WerScorer:
    def __init__(self, cfg=None):
        self.cfg = cfg
        self.reset()

    def reset(self):
        self.distance = 0
        self.ref_length = 0

    def add_string(self, ref: str, pred: str):
        ref_items, pred_items = ref.split(), pred.split()
        self.distance += edit_distance(ref_items, pred_items)
        self.ref_length += len(ref_items)

    def result_string(self) -> str:
        return f"WER: {self.score():.2f}"

    def score(self) -> float:
        return 100.0 * self.distance / self.ref_length if self.ref_length > 0 else 0


def build_scorer(choice, tgt_dict=None):
    """fairseq/scoring/__init__.py:39-48 for the scorer this path uses."""
    name = getattr(choice, "_name", choice)
    if name != "wer":
        raise ValueError(f"scorer {name!r} is not part of this path (the mtl generator scores with 'wer')")
    return WerScorer()
