"""Stand-in for the third-party ``editdistance`` package (absent from this image, un-pinned by the reference), so that
fairseq/scoring/wer.py -- which the reference's mtl generator builds on every call (speech_generator_for_s2st_mtl.py:63) --
can be constructed when oracle/gen_golden_infer_mtl.py runs the reference.  TEST INFRASTRUCTURE (build container only).
``eval(a, b)`` = Levenshtein distance (unit-cost insert / delete / substitute), the function the package documents.  The
WER NUMBER in the golden therefore rests on this restatement ("parity unpinned" for the distance itself); the hypothesis
strings and token ids it is computed from are the reference's own."""


def eval(a, b):  # noqa: A001  (the package's API name)
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, y in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y))
        prev = cur
    return prev[len(b)]


distance = eval
