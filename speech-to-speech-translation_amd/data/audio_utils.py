"""Feature / waveform access of the on-disk data path.

Counterpart of ``fairseq/data/audio/audio_utils.py:175-215`` (``parse_path``, ``read_from_stored_zip``: byte-range
reads out of an UNCOMPRESSED zip, addressed as ``<zip path>:<byte offset>:<byte length>``) and
``fairseq/data/audio/speech_to_text_dataset.py:40-94`` (``get_features_or_waveform``).  Pre-extracted ``.npy``
features are the training path's input; waveforms (HuBERT mode) are read with ``soundfile`` when it is installed,
else PCM ``.wav`` through ``scipy.io.wavfile`` (same samples, scaled to [-1, 1) like ``soundfile`` does).
"""
from __future__ import annotations

import io
import mmap
from pathlib import Path
from typing import List, Tuple

import numpy as np

FEATURE_OR_SF_AUDIO_FILE_EXTENSIONS = {".npy", ".wav", ".flac", ".ogg"}


def parse_path(path: str) -> Tuple[str, List[int]]:
    if Path(path).suffix in FEATURE_OR_SF_AUDIO_FILE_EXTENSIONS:
        return path, []
    _path, *slice_ptr = path.split(":")
    if not Path(_path).is_file():
        raise FileNotFoundError(f"File not found: {_path}")
    assert len(slice_ptr) in {0, 2}, f"Invalid path: {path}"
    return _path, [int(i) for i in slice_ptr]


def read_from_stored_zip(zip_path: str, offset: int, length: int) -> bytes:
    with open(zip_path, "rb") as f:
        with mmap.mmap(f.fileno(), length=0, access=mmap.ACCESS_READ) as m:
            return m[offset:offset + length]


def is_npy_data(data: bytes) -> bool:
    return data[0] == 147 and data[1] == 78


def is_sf_audio_data(data: bytes) -> bool:
    return data[:3] in (b"RIF", b"fLa", b"Ogg")


def read_waveform(path_or_fp) -> Tuple[np.ndarray, int]:
    """float32 waveform in [-1, 1) and its sample rate."""
    try:
        import soundfile as sf
        wav, sr = sf.read(path_or_fp, dtype="float32", always_2d=False)
        return wav, sr
    except ImportError:
        from scipy.io import wavfile
        sr, wav = wavfile.read(path_or_fp)
        if wav.dtype == np.int16:
            wav = wav.astype(np.float32) / 32768.0
        elif wav.dtype == np.int32:
            wav = wav.astype(np.float32) / 2147483648.0
        elif wav.dtype == np.uint8:
            wav = (wav.astype(np.float32) - 128.0) / 128.0
        else:
            wav = wav.astype(np.float32)
        return wav, sr


def get_features_or_waveform(path: str, need_waveform: bool = False, use_sample_rate=None) -> np.ndarray:
    _path, slice_ptr = parse_path(path)
    if len(slice_ptr) == 0:
        ext = Path(_path).suffix
        if ext not in FEATURE_OR_SF_AUDIO_FILE_EXTENSIONS:
            raise ValueError(f'Unsupported file format for "{_path}"')
        if need_waveform:
            return read_waveform(_path)[0]
        if ext != ".npy":
            raise NotImplementedError("on-the-fly fbank extraction from audio is outside the hot path: "
                                      "pre-extract features to .npy / zip as the recipe does")
        return np.load(_path)
    assert _path.endswith(".zip")
    data = read_from_stored_zip(_path, slice_ptr[0], slice_ptr[1])
    if is_npy_data(data):
        return np.load(io.BytesIO(data))
    if is_sf_audio_data(data) and need_waveform:
        return read_waveform(io.BytesIO(data))[0]
    raise ValueError(f'Unknown file format for "{path}"')
