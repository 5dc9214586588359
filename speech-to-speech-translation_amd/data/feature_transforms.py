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
        if time_warp_w > 0:
            raise NotImplementedError("time warping needs OpenCV (cv2.resize) in the reference; not available here")
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
