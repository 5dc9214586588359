"""Feature transforms of the on-disk data path, same registry names and config keys as
``fairseq/data/audio/feature_transforms/`` (``__init__.py:14-106``, ``utterance_cmvn.py``, ``global_cmvn.py`` with the
fork's ``src_global_cmvn`` / ``tgt_global_cmvn`` twins, ``specaugment.py``).  They run on the host, per utterance,
in numpy -- the same operations in the same order, so the results are bit-identical; SpecAugment draws from
numpy's global RNG exactly as the reference does.
"""
from __future__ import annotations

import math
import numbers
from typing import Dict, Optional

import numpy as np

AUDIO_FEATURE_TRANSFORM_REGISTRY: Dict[str, type] = {}


def register_audio_feature_transform(name):
    def deco(cls):
        if name in AUDIO_FEATURE_TRANSFORM_REGISTRY:
            raise ValueError(f"Cannot register duplicate transform ({name})")
        AUDIO_FEATURE_TRANSFORM_REGISTRY[name] = cls
        return cls
    return deco


def get_audio_feature_transform(name):
    return AUDIO_FEATURE_TRANSFORM_REGISTRY[name]


@register_audio_feature_transform("utterance_cmvn")
class UtteranceCMVN:
    @classmethod
    def from_config_dict(cls, config=None):
        c = {} if config is None else config
        return cls(c.get("norm_means", True), c.get("norm_vars", True))

    def __init__(self, norm_means=True, norm_vars=True):
        self.norm_means, self.norm_vars = norm_means, norm_vars

    def __call__(self, x):
        mean = x.mean(axis=0)
        square_sums = (x ** 2).sum(axis=0)
        if self.norm_means:
            x = np.subtract(x, mean)
        if self.norm_vars:
            var = square_sums / x.shape[0] - mean ** 2
            std = np.sqrt(np.maximum(var, 1e-10))
            x = np.divide(x, std)
        return x


class _GlobalCMVN:
    @classmethod
    def from_config_dict(cls, config=None):
        return cls(({} if config is None else config).get("stats_npz_path"))

    def __init__(self, stats_npz_path):
        self.stats_npz_path = stats_npz_path
        stats = np.load(stats_npz_path)
        self.mean, self.std = stats["mean"], stats["std"]

    def __call__(self, x):
        return np.divide(np.subtract(x, self.mean), self.std)


@register_audio_feature_transform("global_cmvn")
class GlobalCMVN(_GlobalCMVN):
    pass


@register_audio_feature_transform("src_global_cmvn")
class SRCGlobalCMVN(_GlobalCMVN):
    pass


@register_audio_feature_transform("tgt_global_cmvn")
class TGTGlobalCMVN(_GlobalCMVN):
    pass


def resize_rows_linear(src: np.ndarray, new_rows: int) -> np.ndarray:
    """``cv2.resize(src, dsize=(src.shape[1], new_rows), interpolation=cv2.INTER_LINEAR)`` for a 2-D float array whose
    width stays (what the time warp of specaugment.py:100-109 asks OpenCV for).  OpenCV's bilinear resize, restated: output
    row y samples the source at fy = (y + 0.5) * (rows / new_rows) - 0.5 (pixel centres aligned, the scale in double, fy in
    float), sy = floor(fy), weights (1 - (fy - sy), fy - sy) in float32 on rows sy and sy + 1, each index clamped into the
    image; the horizontal pass has scale 1 and weights (1, 0), i.e. it is the identity; equal sizes are a plain copy.
    OpenCV is not part of this image: checked against a float64 restatement and the operator's invariants
    (``tests/test_data_audio.py``), not against cv2 itself."""
    rows = src.shape[0]
    if new_rows == rows:
        return src.copy()
    scale = 1.0 / (new_rows / float(rows))
    fy = ((np.arange(new_rows, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    sy = np.floor(fy).astype(np.int64)
    fy = (fy - sy.astype(np.float32)).astype(np.float32)
    i0 = np.clip(sy, 0, rows - 1)
    i1 = np.clip(sy + 1, 0, rows - 1)
    if src.dtype == np.float64:
        b0, b1 = (1.0 - fy.astype(np.float64))[:, None], fy.astype(np.float64)[:, None]
    else:
        b0, b1 = (np.float32(1.0) - fy)[:, None], fy[:, None]
    return (src[i0] * b0 + src[i1] * b1).astype(src.dtype)


@register_audio_feature_transform("specaugment")
class SpecAugmentTransform:
    @classmethod
    def from_config_dict(cls, config=None):
        c = {} if config is None else config
        return cls(c.get("time_warp_W", 0), c.get("freq_mask_N", 0), c.get("freq_mask_F", 0), c.get("time_mask_N", 0),
                   c.get("time_mask_T", 0), c.get("time_mask_p", 0.0), c.get("mask_value", None))

    def __init__(self, time_warp_w=0, freq_mask_n=0, freq_mask_f=0, time_mask_n=0, time_mask_t=0, time_mask_p=0.0,
                 mask_value: Optional[float] = 0.0):
        assert mask_value is None or isinstance(mask_value, numbers.Number)
        if freq_mask_n > 0:
            assert freq_mask_f > 0
        if time_mask_n > 0:
            assert time_mask_t > 0
        self.time_warp_w = time_warp_w
        self.freq_mask_n, self.freq_mask_f = freq_mask_n, freq_mask_f
        self.time_mask_n, self.time_mask_t, self.time_mask_p = time_mask_n, time_mask_t, time_mask_p
        self.mask_value = mask_value

    def __call__(self, spectrogram):
        assert len(spectrogram.shape) == 2
        distorted = spectrogram.copy()
        num_frames, num_freqs = spectrogram.shape
        mask_value = self.mask_value
        if mask_value is None:
            mask_value = spectrogram.mean()
        if num_frames == 0 or num_freqs < self.freq_mask_f:
            return spectrogram
        if self.time_warp_w > 0 and 2 * self.time_warp_w < num_frames:
            # specaugment.py:95-110: a random frame w0 moves by w; the parts before / after it are stretched to fit
            # (two draws, in this order, before the masks' draws)
            w0 = np.random.randint(self.time_warp_w, num_frames - self.time_warp_w)
            w = np.random.randint(-self.time_warp_w + 1, self.time_warp_w)
            distorted = np.concatenate((resize_rows_linear(distorted[:w0, :], w0 + w),
                                        resize_rows_linear(distorted[w0:, :], num_frames - w0 - w)), axis=0)
        for _ in range(self.freq_mask_n):  # two draws per mask, in this order (specaugment.py:110-114)
            f = np.random.randint(0, self.freq_mask_f)
            f0 = np.random.randint(0, num_freqs - f)
            if f != 0:
                distorted[:, f0:f0 + f] = mask_value
        max_t = min(self.time_mask_t, math.floor(num_frames * self.time_mask_p))
        if max_t < 1:
            return distorted
        for _ in range(self.time_mask_n):
            t = np.random.randint(0, max_t)
            t0 = np.random.randint(0, num_frames - t)
            if t != 0:
                distorted[t0:t0 + t, :] = mask_value
        return distorted


class CompositeAudioFeatureTransform:
    """``key`` selects the fork's per-side lists: "transforms", "src_transforms" or "tgt_transforms"."""

    @classmethod
    def from_config_dict(cls, config=None, key: str = "transforms"):
        c = {} if config is None else config
        names = c.get(key)
        if names is None:
            return None
        return cls([get_audio_feature_transform(t).from_config_dict(c.get(t)) for t in names])

    def __init__(self, transforms):
        self.transforms = [t for t in transforms if t is not None]

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x
