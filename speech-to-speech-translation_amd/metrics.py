"""MCD evaluation on the HIP path: counterpart of ``batch_dynamic_time_warping`` and
``batch_mel_cepstral_distortion`` (examples/s2s_trans/tasks/s2s_translation.py:414-552).

DTW runs as one workgroup per pair over anti-diagonals; the MFCC front end is dense-DFT / mel / DCT
GEMMs (bf16x3 precise mode) around two glue kernels.  The reference takes MFCC from torchaudio
(absent in this image, un-pinned): ``MFCC`` restates torchaudio.transforms.MFCC's documented
defaults (power-2 centred reflect-padded STFT with a periodic Hann window, HTK mel filterbank without
normalisation, log(mel + 1e-6), orthonormal DCT-II) -- parity un-pinned for that transform; the DTW
is pinned by goldens from the reference function.
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch

from .runtime import binding as bd


def batch_dynamic_time_warping(distance: torch.Tensor, shapes: Optional[torch.Tensor] = None):
    """distance [B, M, N] (device) -> (cumdist, backptr int32, pathmap int32), as the reference."""
    bd.require_device(distance)
    d = distance.to(torch.float32).contiguous()
    B, M, N = d.shape
    cum = torch.empty_like(d)
    bp = torch.empty(B, M, N, dtype=torch.int32, device=d.device)
    pm = torch.empty(B, M, N, dtype=torch.int32, device=d.device)
    sh = shapes.to(d.device, torch.int32).contiguous() if shapes is not None else None
    bd.call("s2st_dtw_f32", d, sh, B, M, N, cum, bp, pm)
    return cum, bp, pm


class MFCC:
    def __init__(self, sample_rate: int, device, n_mfcc: int = 13, n_mels: int = 80, f_min: float = 20.0):
        self.sample_rate, self.device, self.n_mfcc, self.n_mels = sample_rate, device, n_mfcc, n_mels
        self.n_fft = self.win = int(0.05 * sample_rate)
        self.hop = int(0.0125 * sample_rate)
        self.F = self.n_fft // 2 + 1
        win = torch.hann_window(self.win, periodic=True, dtype=torch.float64)
        k = torch.arange(self.F, dtype=torch.float64).unsqueeze(1)
        n = torch.arange(self.n_fft, dtype=torch.float64).unsqueeze(0)
        ang = 2 * math.pi * k * n / self.n_fft
        basis = torch.cat([torch.cos(ang), -torch.sin(ang)], 0) * win  # [2F][n_fft]
        self.basis = basis.float().contiguous().to(device)
        # HTK mel filterbank, norm=None (torchaudio.functional.melscale_fbanks)
        all_freqs = torch.linspace(0, sample_rate // 2, self.F, dtype=torch.float64)
        hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
        m_pts = torch.linspace(hz2mel(f_min), hz2mel(sample_rate / 2.0), n_mels + 2, dtype=torch.float64)
        f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
        f_diff = f_pts[1:] - f_pts[:-1]
        slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
        down = -slopes[:, :-2] / f_diff[:-1]
        up = slopes[:, 2:] / f_diff[1:]
        fb = torch.clamp(torch.min(down, up), min=0.0)  # [F][n_mels]
        self.fb_t = fb.t().float().contiguous().to(device)  # [n_mels][F]: K-contiguous GEMM operand
        # orthonormal DCT-II (torchaudio.functional.create_dct)
        nn_ = torch.arange(n_mels, dtype=torch.float64)
        kk = torch.arange(n_mfcc, dtype=torch.float64).unsqueeze(1)
        dct = torch.cos(math.pi / n_mels * (nn_ + 0.5) * kk)
        dct[0] *= 1.0 / math.sqrt(2.0)
        dct *= math.sqrt(2.0 / n_mels)
        self.dct = dct.float().contiguous().to(device)  # [n_mfcc][n_mels]

    def __call__(self, y: torch.Tensor) -> torch.Tensor:
        """waveform [N] -> MFCC [T, n_mfcc] (the reference transposes torchaudio's [n_mfcc, T])."""
        dev = self.device
        y = y.to(dev, torch.float32).contiguous()
        n = y.numel()
        T = 1 + n // self.hop
        padded = torch.empty(n + self.n_fft, device=dev)
        bd.call("s2st_reflect_pad_f32", y, padded, n, self.n_fft // 2)
        Y = torch.empty(T, 2 * self.F, device=dev)
        bd.gemm(padded, self.basis, Y, T, 2 * self.F, self.n_fft, a_ld=self.hop, precise=True)
        P = torch.empty(T, self.F, device=dev)
        bd.call("s2st_power_spec_f32", Y, P, T, self.F)
        mel = torch.empty(T, self.n_mels, device=dev)
        bd.gemm(P, self.fb_t, mel, T, self.n_mels, self.F, precise=True)
        bd.call("s2st_log_offset_f32", mel, T * self.n_mels, 1e-6)
        out = torch.empty(T, self.n_mfcc, device=dev)
        bd.gemm(mel, self.dct, out, T, self.n_mfcc, self.n_mels, precise=True)
        return out


def batch_mel_cepstral_distortion(y1: List[torch.Tensor], y2: List[torch.Tensor], sr: int, normalize_type: str = "path",
                                  mfcc_fn: Optional[MFCC] = None, device=None):
    """Returns [(distortion, (x1, x2, dist, cumdist, backptr, pathmap)), ...] like the reference."""
    device = device or (mfcc_fn.device if mfcc_fn is not None else y1[0].device)
    if mfcc_fn is None or mfcc_fn.sample_rate != sr:
        mfcc_fn = MFCC(sr, device)
    x1 = [mfcc_fn(a) for a in y1]
    x2 = [mfcc_fn(b) for b in y2]
    max_m, max_n = max(a.shape[0] for a in x1), max(b.shape[0] for b in x2)
    d = torch.zeros(len(x1), max_m, max_n, device=device)
    for b, (a, c) in enumerate(zip(x1, x2)):
        bd.call("s2st_rms_dist_f32", a, c, d[b], a.shape[0], c.shape[0], a.shape[1], max_n)
    s = torch.tensor([[a.shape[0], c.shape[0]] for a, c in zip(x1, x2)], dtype=torch.int32)
    cum, bp, pm = batch_dynamic_time_warping(d, s)
    rets = []
    for b, (m, n) in enumerate(s.tolist()):
        cumdist, backptr, pathmap = cum[b, :m, :n], bp[b, :m, :n], pm[b, :m, :n]
        div = {None: 1, "len1": m, "len2": n}.get(normalize_type)
        if div is None:
            if normalize_type != "path":
                raise ValueError(f"normalize_type {normalize_type} not supported")
            div = int(pathmap.sum())
        rets.append((cumdist[-1, -1] / div, (x1[b], x2[b], d[b, :m, :n], cumdist, backptr, pathmap)))
    return rets
