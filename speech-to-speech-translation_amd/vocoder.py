"""Griffin-Lim vocoder on the HIP path: the counterpart of
``fairseq/models/text_to_speech/vocoder.py:24-158`` (PseudoInverseMelScale, GriffinLim,
GriffinLimVocoder) with the same constructor arguments.

Like the reference, the STFT / inverse STFT are dense-DFT contractions (the reference uses conv1d /
conv_transpose1d with Fourier bases, audio_utils.py:259-271, vocoder.py:56-98): here they are GEMMs on
the matrix cores (bf16x3 "precise" mode: phase retrieval is precision-sensitive) around small HIP
kernels for polar <-> rectangular conversion, reflect padding and overlap-add.  The constant tables
(window, Fourier bases and their pseudo-inverses, mel filterbank pseudo-inverse) are built once on
the host at construction, as the reference does in its ``register_buffer`` calls.

The mel filterbank comes from librosa in the reference (absent in this image, version un-pinned):
``slaney_mel_filters`` restates librosa.filters.mel's defaults (htk=False, norm='slaney').
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .runtime import binding as bd


def get_window(n_fft: int, win_length: int, window_fn=torch.hann_window) -> torch.Tensor:
    padding = n_fft - win_length
    assert padding >= 0
    return F.pad(window_fn(win_length), (padding // 2, padding - padding // 2))


def get_fourier_basis(n_fft: int) -> torch.Tensor:
    basis = np.fft.fft(np.eye(n_fft))
    basis = np.vstack([np.real(basis[:n_fft // 2 + 1, :]), np.imag(basis[:n_fft // 2 + 1, :])])
    return torch.from_numpy(basis).float()


def slaney_mel_filters(sample_rate: int, n_fft: int, n_mels: int, f_min: float, f_max: float) -> torch.Tensor:
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    f_max = sample_rate / 2.0 if f_max is None else f_max
    fft_f = np.linspace(0, sample_rate / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return torch.from_numpy(w.astype(np.float32))


class GriffinLim:
    def __init__(self, n_fft: int, win_length: int, hop_length: int, n_iter: int, device, window_fn=torch.hann_window):
        self.n_fft, self.win_length, self.hop_length, self.n_iter, self.device = n_fft, win_length, hop_length, n_iter, device
        win = get_window(n_fft, win_length, window_fn)
        self.F = n_fft // 2 + 1
        # forward basis [2F][n_fft] (TTSSpectrogram) and inverse basis [2F][n_fft] (GriffinLim.__init__)
        self.fwd = (get_fourier_basis(n_fft) * win).contiguous().to(device)
        inv = torch.pinverse(n_fft / hop_length * get_fourier_basis(n_fft)).T * win
        self.inv_t = inv.t().contiguous().to(device)  # [n_fft][2F]: rows-of-output x K layout for the GEMM
        self.win_sq = win ** 2
        self._wss = {}

    def _window_sum_square(self, n_frames: int) -> torch.Tensor:
        w = self._wss.get(n_frames)
        if w is None:  # vocoder.py:69-80 (a constant of the frame count)
            n = self.n_fft + self.hop_length * (n_frames - 1)
            x = torch.zeros(n, dtype=torch.float32)
            for i in range(n_frames):
                o = i * self.hop_length
                x[o: min(n, o + self.n_fft)] += self.win_sq[:max(0, min(self.n_fft, n - o))]
            w = self._wss[n_frames] = x.to(self.device)
        return w

    def _inverse(self, X: torch.Tensor, T: int) -> torch.Tensor:
        """X [T][2F] (real | imag) -> waveform [hop * (T - 1)] (vocoder.py:82-98)."""
        frames = torch.empty(T, self.n_fft, device=self.device)
        bd.gemm(X, self.inv_t, frames, T, self.n_fft, 2 * self.F, precise=True)
        n_out = self.hop_length * (T - 1)
        wave = torch.empty(n_out, device=self.device)
        bd.call("s2st_gl_overlap_add_f32", frames, self._window_sum_square(T), wave, T, self.n_fft, self.hop_length, n_out)
        return wave

    def _transform(self, wave: torch.Tensor, T: int) -> torch.Tensor:
        """waveform -> STFT [T][2F] (audio_utils.py:259-271: reflect pad + strided Fourier-basis conv)."""
        n = wave.numel()
        padded = torch.empty(n + self.n_fft, device=self.device)
        bd.call("s2st_reflect_pad_f32", wave, padded, n, self.n_fft // 2)
        Y = torch.empty(T, 2 * self.F, device=self.device)
        bd.gemm(padded, self.fwd, Y, T, 2 * self.F, self.n_fft, a_ld=self.hop_length, precise=True)
        return Y

    def __call__(self, specgram: torch.Tensor, angles: np.ndarray = None) -> torch.Tensor:
        """specgram [F, T] magnitudes -> waveform.  ``angles`` defaults to the reference's draw
        np.angle(np.exp(2j*pi*np.random.rand(F, T))) from numpy's global RNG (vocoder.py:101-102)."""
        Fq, T = specgram.shape
        assert Fq == self.F
        if angles is None:
            angles = np.angle(np.exp(2j * np.pi * np.random.rand(Fq, T)))
        mag = specgram.to(self.device, torch.float32).contiguous()
        ang = torch.from_numpy(np.ascontiguousarray(angles, dtype=np.float32)).to(self.device)
        X = torch.empty(T, 2 * Fq, device=self.device)
        bd.call("s2st_gl_polar_f32", mag, ang, X, Fq, T)
        wave = self._inverse(X, T)
        for _ in range(self.n_iter):
            Y = self._transform(wave, T)
            bd.call("s2st_gl_project_f32", mag, Y, X, Fq, T)
            wave = self._inverse(X, T)
        return wave


class GriffinLimVocoder:
    def __init__(self, sample_rate, win_size, hop_size, n_fft, n_mels, f_min, f_max, window_fn=torch.hann_window,
                 spec_bwd_max_iter=32, device=None):
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.n_mels, self.F = n_mels, n_fft // 2 + 1
        basis = torch.pinverse(slaney_mel_filters(sample_rate, n_fft, n_mels, f_min, f_max))  # F x n_mels
        self.inv_mel = basis.contiguous().to(self.device)
        self.gl = GriffinLim(n_fft, win_size, hop_size, spec_bwd_max_iter, self.device, window_fn)

    def __call__(self, x: torch.Tensor, angles: np.ndarray = None) -> torch.Tensor:
        """x [T, n_mels] log-mel -> waveform [1, N] (vocoder.py:136-144)."""
        T, C_ = x.shape
        xt = torch.empty(C_, T, device=self.device)  # exp(x)^T
        bd.call("s2st_exp_transpose_f32", x.to(self.device, torch.float32).contiguous(), xt, T, C_)
        spec = torch.empty(self.F, T, device=self.device)
        bd.gemm(self.inv_mel, xt, spec, self.F, T, C_, b_kmajor=False, b_ld=T, precise=True)
        bd.call("s2st_clamp_min_f32", spec, self.F * T, 0.0)
        return self.gl(spec, angles).unsqueeze(0)

    @classmethod
    def from_data_cfg(cls, args, data_cfg, device=None):
        feat_cfg = data_cfg.config["features"] if hasattr(data_cfg, "config") else data_cfg["features"]
        return cls(sample_rate=feat_cfg["sample_rate"], win_size=int(feat_cfg["win_len_t"] * feat_cfg["sample_rate"]),
                   hop_size=int(feat_cfg["hop_len_t"] * feat_cfg["sample_rate"]), n_fft=feat_cfg["n_fft"],
                   n_mels=feat_cfg["n_mels"], f_min=feat_cfg["f_min"], f_max=feat_cfg["f_max"],
                   window_fn=getattr(torch, feat_cfg["window_fn"] + "_window"),
                   spec_bwd_max_iter=getattr(args, "spec_bwd_max_iter", 32), device=device)
