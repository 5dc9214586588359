"""CPU restatement of the data path's integer work -- TEST INFRASTRUCTURE (only tests/ may import it).

``batch_by_size``: behavioural restatement of ``batch_by_size_vec`` (fairseq/data/data_utils_fast.pyx:20-100).
Pinned against the reference's own Cython extension built into oracle/_ref by oracle/build_ref.sh (when the
reference is present) and against tests/golden/batcher.npz generated from that build (oracle/gen_golden_data.py).

``kaldi_fbank_f64`` / ``resize_rows_linear_f64`` (end of the file): PARITY UNPINNED -- they restate torchaudio's Kaldi
filter bank and OpenCV's bilinear resize, neither of which is installed here or vendored by the reference.
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




# ---- on-the-fly features and the SpecAugment time warp: restatements of two ABSENT third-party packages -------------------
# PARITY UNPINNED for the two functions below: torchaudio (fairseq/data/audio/audio_utils.py:131-145 ->
# torchaudio.compliance.kaldi.fbank) and OpenCV (fairseq/data/audio/feature_transforms/specaugment.py:95-110 -> cv2.resize)
# are not installed in this image and the reference holds no fixture for either.  They restate the packages' published
# algorithms in float64, sample by sample, independently of the vectorised float32 product code they check.


def kaldi_fbank_f64(waveform: np.ndarray, sample_rate: float, n_bins: int = 80) -> np.ndarray:
    """Kaldi's FbankComputer with torchaudio's defaults (frame 25 ms / shift 10 ms, snip_edges, no dither, remove DC offset,
    pre-emphasis 0.97, povey window, power spectrum of the zero-padded frame, triangular mel filters from 20 Hz to Nyquist in
    mel = 1127 ln(1 + f / 700), log floored at float32 epsilon), one frame and one filter at a time."""
    import math
    x = np.asarray(waveform, dtype=np.float64)
    x = x[0] if x.ndim == 2 else x
    shift, size = int(sample_rate * 0.01), int(sample_rate * 0.025)
    padded = 1 << (size - 1).bit_length()
    if len(x) < size:
        return np.zeros((0, n_bins))
    n_frames = 1 + (len(x) - size) // shift
    win = [(0.5 - 0.5 * math.cos(2.0 * math.pi * i / (size - 1))) ** 0.85 for i in range(size)]
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    lo, hi = mel(20.0), mel(0.5 * sample_rate)
    d = (hi - lo) / (n_bins + 1)
    out = np.zeros((n_frames, n_bins))
    for t in range(n_frames):
        fr = x[t * shift:t * shift + size].copy()
        fr -= fr.sum() / size
        pre = fr.copy()
        for i in range(size - 1, 0, -1):
            pre[i] = fr[i] - 0.97 * fr[i - 1]
        pre[0] = fr[0] - 0.97 * fr[0]
        buf = np.zeros(padded)
        buf[:size] = pre * win
        sp = np.fft.rfft(buf)
        pw = sp.real ** 2 + sp.imag ** 2
        for b in range(n_bins):
            left, center, right = lo + b * d, lo + (b + 1) * d, lo + (b + 2) * d
            e = 0.0
            for k in range(padded // 2):  # (the Nyquist bin carries no weight)
                m = mel(k * sample_rate / padded)
                if left < m < right:
                    e += pw[k] * ((m - left) / (center - left) if m <= center else (right - m) / (right - center))
            out[t, b] = math.log(max(e, float(np.finfo(np.float32).eps)))
    return out


def resize_rows_linear_f64(src: np.ndarray, new_rows: int) -> np.ndarray:
    """Bilinear resize along the rows with aligned pixel centres (OpenCV's INTER_LINEAR convention), row by row."""
    import math
    src = np.asarray(src, dtype=np.float64)
    rows = src.shape[0]
    out = np.zeros((new_rows, src.shape[1]))
    for y in range(new_rows):
        f = (y + 0.5) * rows / new_rows - 0.5
        s = math.floor(f)
        a = f - s
        r0, r1 = min(max(s, 0), rows - 1), min(max(s + 1, 0), rows - 1)
        out[y] = (1.0 - a) * src[r0] + a * src[r1]
    return out
