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


def get_waveform(path_or_fp, normalization: bool = True, mono: bool = True, always_2d: bool = True,
                 output_sample_rate=None) -> Tuple[np.ndarray, int]:
    """``fairseq/data/audio/audio_utils.py:65-108``: channels x samples, float32; ``normalization=False`` scales to the
    16-bit range the Kaldi front end expects.  ``mono`` mixes channels down by their mean (what sox's ``channels 1`` does,
    ``:48-50``).  A sample-rate change is sox's ``rate`` effect there (a polyphase resampler this package does not restate):
    asking for one that differs from the file's is refused."""
    if isinstance(path_or_fp, str) and Path(path_or_fp).suffix not in {".wav", ".flac", ".ogg"}:
        raise ValueError(f"Unsupported audio format: {Path(path_or_fp).suffix}")
    wav, sr = read_waveform(path_or_fp)
    wav = np.asarray(wav, dtype=np.float32)
    wav = wav[None, :] if wav.ndim == 1 else wav.T  # T x C -> C x T
    if output_sample_rate is not None and int(output_sample_rate) != int(sr):
        raise NotImplementedError(f"resampling {sr} -> {output_sample_rate} Hz (sox 'rate' in the reference) is not available: "
                                  "store the audio at the sample rate the model expects")
    if mono and wav.shape[0] > 1:
        wav = wav.mean(axis=0, keepdims=True, dtype=np.float32)
    if not normalization:
        wav = wav * np.float32(2 ** 15)
    if not always_2d:
        wav = wav.squeeze(axis=0)
    return wav, sr


def _mel_scale(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


_FBANK_TABLES: dict = {}


def _kaldi_tables(sample_rate: float, n_bins: int, frame_length_ms: float, frame_shift_ms: float, low_freq: float,
                  high_freq: float):
    """Window and mel filter bank of ``torchaudio.compliance.kaldi.fbank`` (Kaldi's ``FbankComputer``): povey window =
    symmetric Hann ** 0.85; triangular filters equally spaced on mel = 1127 ln(1 + f / 700) between ``low_freq`` and
    Nyquist + ``high_freq``, evaluated at the FFT bins' mel values (no Nyquist bin: its column is zero)."""
    key = (float(sample_rate), n_bins, frame_length_ms, frame_shift_ms, low_freq, high_freq)
    t = _FBANK_TABLES.get(key)
    if t is not None:
        return t
    shift = int(sample_rate * 0.001 * frame_shift_ms)
    size = int(sample_rate * 0.001 * frame_length_ms)
    padded = 1
    while padded < size:
        padded *= 2
    n = np.arange(size, dtype=np.float32)
    hann = (np.float32(0.5) - np.float32(0.5) * np.cos(np.float32(2.0 * np.pi) * n / np.float32(size - 1))).astype(np.float32)
    window = np.power(hann, np.float32(0.85)).astype(np.float32)
    nbin_fft = padded // 2
    nyquist = 0.5 * sample_rate
    hi = high_freq + nyquist if high_freq <= 0.0 else high_freq
    assert 0.0 <= low_freq < nyquist and 0.0 < hi <= nyquist and low_freq < hi
    width = sample_rate / padded
    mlo, mhi = _mel_scale(low_freq), _mel_scale(hi)
    delta = np.float32((mhi - mlo) / (n_bins + 1))
    b = np.arange(n_bins, dtype=np.float32)[:, None]
    left = np.float32(mlo) + b * delta
    center = np.float32(mlo) + (b + np.float32(1.0)) * delta
    right = np.float32(mlo) + (b + np.float32(2.0)) * delta
    mel = _mel_scale((np.float32(width) * np.arange(nbin_fft, dtype=np.float32)).astype(np.float32)).astype(np.float32)[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    banks = np.maximum(np.float32(0.0), np.minimum(up, down)).astype(np.float32)
    banks = np.concatenate([banks, np.zeros((n_bins, 1), np.float32)], axis=1)  # [n_bins, padded / 2 + 1]
    t = _FBANK_TABLES[key] = (shift, size, padded, window, np.ascontiguousarray(banks.T))
    return t


def kaldi_fbank(waveform: np.ndarray, sample_rate: float, n_bins: int = 80) -> np.ndarray:
    """Log mel filter-bank features as ``fairseq/data/audio/audio_utils.py:131-145`` asks ``torchaudio.compliance.kaldi.fbank
    (waveform, num_mel_bins=n_bins, sample_frequency=sample_rate)`` for them -- every other option at its default there:
    channel 0, 25 ms frames every 10 ms, ``snip_edges`` (1 + (N - 400) // 160 frames at 16 kHz), no dither, DC offset removed
    per frame, pre-emphasis 0.97 (first sample against itself), povey window, zero-padded to the next power of two, POWER
    spectrum, mel banks from 20 Hz to Nyquist, ``log(max(e, float32 eps))``, no energy column, no mean subtraction.
    float32 arithmetic like the tensor code there.  ``waveform``: channels x samples in the 16-bit range.
    The package this restates is not part of this image: checked against an independent float64 restatement and Kaldi's
    documented properties (``tests/test_data_audio.py``), not against torchaudio's own output."""
    wav = np.asarray(waveform, dtype=np.float32)
    if wav.ndim == 1:
        wav = wav[None, :]
    wav = wav[0]
    shift, size, padded, window, banks_t = _kaldi_tables(float(sample_rate), n_bins, 25.0, 10.0, 20.0, 0.0)
    if wav.shape[0] < size:
        return np.zeros((0, n_bins), np.float32)
    m = 1 + (wav.shape[0] - size) // shift
    frames = np.lib.stride_tricks.as_strided(wav, shape=(m, size), strides=(shift * wav.strides[0], wav.strides[0]))
    x = frames - frames.mean(axis=1, keepdims=True, dtype=np.float32)
    prev = np.concatenate([x[:, :1], x[:, :-1]], axis=1)
    x = (x - np.float32(0.97) * prev) * window[None, :]
    if padded != size:
        x = np.concatenate([x, np.zeros((m, padded - size), np.float32)], axis=1)
    spec = np.fft.rfft(x.astype(np.float32), axis=1)
    power = (spec.real.astype(np.float32) ** 2 + spec.imag.astype(np.float32) ** 2).astype(np.float32)
    e = power @ banks_t
    return np.log(np.maximum(e, np.finfo(np.float32).eps)).astype(np.float32)


def get_fbank(path_or_fp, n_bins: int = 80) -> np.ndarray:
    """``fairseq/data/audio/audio_utils.py:148-164``: features of an audio file extracted on the fly (the waveform is NOT
    normalised: Kaldi works on the 16-bit range)."""
    wav, sr = get_waveform(path_or_fp, normalization=False)
    return kaldi_fbank(wav, sr, n_bins)


def get_features_or_waveform(path: str, need_waveform: bool = False, use_sample_rate=None) -> np.ndarray:
    """``fairseq/data/audio/speech_to_text_dataset.py:40-96``."""
    _path, slice_ptr = parse_path(path)
    if len(slice_ptr) == 0:
        ext = Path(_path).suffix
        if ext not in FEATURE_OR_SF_AUDIO_FILE_EXTENSIONS:
            raise ValueError(f'Unsupported file format for "{_path}"')
        if need_waveform:
            return get_waveform(_path, always_2d=False, output_sample_rate=use_sample_rate)[0]
        return np.load(_path) if ext == ".npy" else get_fbank(_path)
    assert _path.endswith(".zip")
    data = read_from_stored_zip(_path, slice_ptr[0], slice_ptr[1])
    if is_npy_data(data):
        return npy_from_bytes(data)
    if is_sf_audio_data(data):
        f = io.BytesIO(data)
        return get_waveform(f, always_2d=False, output_sample_rate=use_sample_rate)[0] if need_waveform else get_fbank(f)
    raise ValueError(f'Unknown file format for "{path}"')
