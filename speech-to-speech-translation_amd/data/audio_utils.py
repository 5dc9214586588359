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

import re

import numpy as np

FEATURE_OR_SF_AUDIO_FILE_EXTENSIONS = {".npy", ".wav", ".flac", ".ogg"}


_IS_FILE: dict = {}


def parse_path(path: str) -> Tuple[str, List[int]]:
    if Path(path).suffix in FEATURE_OR_SF_AUDIO_FILE_EXTENSIONS:
        return path, []
    _path, *slice_ptr = path.split(":")
    ok = _IS_FILE.get(_path)
    if ok is None:  # one stat per archive and process, not per item
        ok = _IS_FILE[_path] = Path(_path).is_file()
    if not ok:
        raise FileNotFoundError(f"File not found: {_path}")
    assert len(slice_ptr) in {0, 2}, f"Invalid path: {path}"
    return _path, [int(i) for i in slice_ptr]


_ZIP_MAPS: dict = {}


def read_from_stored_zip(zip_path: str, offset: int, length: int) -> bytes:
    """Byte range of an uncompressed zip (fairseq/data/audio/audio_utils.py:182-189); the archive is mapped once per
    process instead of once per item."""
    m = _ZIP_MAPS.get(zip_path)
    if m is None:
        f = open(zip_path, "rb")
        m = _ZIP_MAPS[zip_path] = (mmap.mmap(f.fileno(), length=0, access=mmap.ACCESS_READ), f)
    return m[0][offset:offset + length]


_NPY_HDR = re.compile(rb"'descr':\s*'([^']+)'.*'fortran_order':\s*(True|False).*'shape':\s*\(([^)]*)\)", re.S)


def npy_from_bytes(data: bytes) -> np.ndarray:
    """``np.load(io.BytesIO(data))`` for plain .npy payloads (format 1.0 - 3.0, no pickled objects) without the
    file-object and ``ast.literal_eval`` overhead of numpy's reader, which dominates at thousands of items per second;
    anything unusual falls back to numpy."""
    try:
        major = data[6]
        hlen, off = (int.from_bytes(data[8:10], "little"), 10) if major == 1 else (int.from_bytes(data[8:12], "little"), 12)
        m = _NPY_HDR.search(data[off:off + hlen])
        if m is None or m.group(1).startswith((b"|O", b"O")):
            raise ValueError
        shape = tuple(int(x) for x in m.group(3).split(b",") if x.strip())
        a = np.frombuffer(data, dtype=np.dtype(m.group(1).decode()), offset=off + hlen,
                          count=int(np.prod(shape)) if shape else 1).reshape(shape, order="F" if m.group(2) == b"True" else "C")
        return a.copy()  # writable, owns its memory (the mapping's bytes object is released)
    except Exception:
        return np.load(io.BytesIO(data))


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
        return npy_from_bytes(data)
    if is_sf_audio_data(data) and need_waveform:
        return read_waveform(io.BytesIO(data))[0]
    raise ValueError(f'Unknown file format for "{path}"')
