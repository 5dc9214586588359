"""CPU restatement (torch fp32 / numpy) of the inference path of config 5 -- TEST INFRASTRUCTURE.

Only tests/, smoke() and bench's cpu_baseline may import this.  Follows (under /root/reference):
  fairseq/speech_generator_for_s2st.py:46-134 (AutoRegressiveSpeechGenerator.generate),
  fairseq/models/text_to_speech/vocoder.py:24-46 (PseudoInverseMelScale), :49-110 (GriffinLim),
  :113-144 (GriffinLimVocoder.forward), fairseq/data/audio/audio_utils.py:218-271
  (get_window, get_fourier_basis, TTSSpectrogram).
Pinned by tests/golden/infer_*.npz generated from the reference classes (oracle/gen_golden_infer.py),
except the mel filterbank: the reference takes it from librosa (absent here, un-pinned version);
``slaney_mel_filters`` restates librosa.filters.mel's documented defaults (htk=False,
norm='slaney') -- PARITY UNPINNED for that one table.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import functools

import numpy as np
import torch
import torch.nn.functional as F


def _decoder_prefix(dec, prev, enc, key_lens):
    """S2STDecoder on the whole prefix with the incremental path's masking (s2st_transformer.py:369-435
    with incremental_state): every row uses position step + 2 (SinusoidalPositionalEmbedding's
    incremental branch ignores padding), while self-attention keys at or beyond ``key_lens`` (the
    utterance's final length once it has finished) are masked through the cached key padding mask."""
    import s2st_oracle as O
    T = prev.shape[1]
    nopad = torch.zeros(prev.shape[0], T, dtype=torch.bool)
    pos = O.positional_embedding(nopad, dec.a.decoder_embed_dim)
    x = dec.prenet(prev) + dec.pos_emb_alpha * pos
    x = x.transpose(0, 1)
    pad = O.lengths_to_padding_mask(key_lens, T)
    self_pad = pad if bool(pad.any()) else None
    enc_pad = enc["encoder_padding_mask"] if bool(enc["encoder_padding_mask"].any()) else None
    fm = O.future_mask(T)
    attn = None
    n = len(dec.transformer_layers)
    for i, layer in enumerate(dec.transformer_layers):
        x, a_ = layer(x, enc["encoder_out"], enc_pad, fm, self_pad, need_attn=(i == n - 1))
        if a_ is not None:
            attn = a_
    attn = attn.mean(dim=0).transpose(2, 1)
    if dec.layer_norm is not None:
        x = dec.layer_norm(x)
    x = x.transpose(0, 1)
    return dec.feat_proj(x), dec.eos_proj(x), attn


@torch.no_grad()
def ar_generate(m, src, src_lens, max_iter: int, eos_prob_threshold: float, n_frames_per_step: int,
                gcmvn: Optional[Dict[str, np.ndarray]] = None, speaker=None) -> List[Dict[str, torch.Tensor]]:
    """Oracle model ``m`` (s2st_oracle.S2STModel, eval mode).  The reference decodes incrementally with
    a key/value cache; because the decoder is causal that equals re-running it on the whole prefix
    (done here) as long as the always-on Prenet dropout is 0 -- with p > 0 the output is random."""
    m.eval()
    enc = m.encoder(src, src_lens, speaker=speaker) if speaker is not None else m.encoder(src, src_lens)
    bsz = src.shape[0]
    out_dim = m.decoder.out_dim
    raw_dim = out_dim // n_frames_per_step
    feat, attn, eos_prob = [], [], []
    finished = torch.zeros(bsz, dtype=torch.bool)
    out_lens = torch.full((bsz,), max_iter, dtype=torch.long)
    prefix = torch.zeros(bsz, 1, out_dim)
    for step in range(max_iter):
        cur_out_lens = out_lens.clone()
        cur_out_lens.masked_fill_(cur_out_lens.eq(max_iter), step + 1)
        # with a speaker the reference's incremental decoder replaces its ONE input frame by the speaker row at every step
        # (s2st_transformer.py:441-444 on the [B, 1, C] tensor the generator hands over): the whole prefix is speaker rows
        # (the t2s decoder has no table: its speaker enters through the encoder's projection only)
        px = (prefix if speaker is None or m.decoder.embed_speaker is None
              else m.decoder.embed_speaker(speaker).expand(-1, prefix.shape[1], -1))
        f_all, eos, a_all = _decoder_prefix(m.decoder, px, enc, cur_out_lens)
        cur_feat = f_all[:, -1:, :]
        cur_eos = torch.sigmoid(eos[:, -1:, :]).squeeze(2)
        feat.append(cur_feat)
        attn.append(a_all[:, :, -1:])
        eos_prob.append(cur_eos)
        cur_finished = cur_eos.squeeze(1) > eos_prob_threshold
        out_lens.masked_fill_((~finished) & cur_finished, step + 1)
        finished = finished | cur_finished
        if int(finished.sum()) == bsz:
            break
        prefix = torch.cat([prefix, cur_feat], dim=1)
    feat = torch.cat(feat, dim=1)
    feat = m.decoder.postnet(feat) + feat
    eos_prob = torch.cat(eos_prob, dim=1)
    attn = torch.cat(attn, dim=2)
    alignment = attn.max(dim=1)[1]
    feat = feat.reshape(bsz, -1, raw_dim)
    if gcmvn is not None:
        feat = feat * torch.from_numpy(gcmvn["std"]).view(1, 1, -1) + torch.from_numpy(gcmvn["mean"]).view(1, 1, -1)
    eos_prob = eos_prob.repeat_interleave(n_frames_per_step, dim=1)
    attn = attn.repeat_interleave(n_frames_per_step, dim=2)
    alignment = alignment.repeat_interleave(n_frames_per_step, dim=1)
    out_lens = out_lens * n_frames_per_step
    return [{"feature": feat[b, :l], "eos_prob": eos_prob[b, :l], "attn": attn[b, :, :l], "alignment": alignment[b, :l]}
            for b, l in zip(range(bsz), out_lens.tolist())]


# ---- Griffin-Lim ---------------------------------------------------------------------------------
def get_window(n_fft: int, win_length: int) -> torch.Tensor:  # audio_utils.py:218-223 (hann)
    padding = n_fft - win_length
    return F.pad(torch.hann_window(win_length), (padding // 2, padding - padding // 2))


def get_fourier_basis(n_fft: int) -> torch.Tensor:  # audio_utils.py:226-231
    basis = np.fft.fft(np.eye(n_fft))
    basis = np.vstack([np.real(basis[:n_fft // 2 + 1, :]), np.imag(basis[:n_fft // 2 + 1, :])])
    return torch.from_numpy(basis).float()


# (the reference builds both bases ONCE, in the constructors named below; cached here per geometry so that the oracle
# timed as a CPU baseline does not pay a 2050 x 2048 pseudo-inverse per Griffin-Lim iteration)
@functools.lru_cache(maxsize=8)
def stft_basis(n_fft, win_length):  # TTSSpectrogram.__init__
    return get_fourier_basis(n_fft) * get_window(n_fft, win_length)  # [2F, n_fft]


@functools.lru_cache(maxsize=8)
def istft_basis(n_fft, win_length, hop_length):  # GriffinLim.__init__
    basis = torch.pinverse(n_fft / hop_length * get_fourier_basis(n_fft)).T
    return basis * get_window(n_fft, win_length)  # [2F, n_fft]


def window_sum_square(n_frames, hop_length, win_length, n_fft) -> torch.Tensor:  # vocoder.py:69-80
    w_sq = get_window(n_fft, win_length) ** 2
    n = n_fft + hop_length * (n_frames - 1)
    x = torch.zeros(n, dtype=torch.float32)
    for i in range(n_frames):
        ofst = i * hop_length
        x[ofst: min(n, ofst + n_fft)] += w_sq[:max(0, min(n_fft, n - ofst))]
    return x


def gl_inverse(mag, phase, n_fft, win_length, hop_length):  # vocoder.py:82-98; mag/phase [1, F, T]
    x = torch.cat([mag * torch.cos(phase), mag * torch.sin(phase)], dim=1)
    x = F.conv_transpose1d(x, istft_basis(n_fft, win_length, hop_length)[:, None, :], stride=hop_length)
    wss = window_sum_square(mag.shape[-1], hop_length, win_length, n_fft)
    nz = wss > 1.1754944e-38
    x[:, :, nz] /= wss[nz]
    x *= n_fft / hop_length
    x = x[:, :, n_fft // 2:]
    x = x[:, :, :-n_fft // 2:]
    return x


def gl_transform(wave, n_fft, win_length, hop_length):  # audio_utils.py:259-271; wave [1, N]
    x = F.pad(wave.unsqueeze(1), (n_fft // 2, n_fft // 2), mode="reflect")
    x = F.conv1d(x, stft_basis(n_fft, win_length)[:, None, :], stride=hop_length)
    re, im = x[:, :n_fft // 2 + 1, :], x[:, n_fft // 2 + 1:, :]
    return torch.sqrt(re ** 2 + im ** 2), torch.atan2(im, re)


def griffin_lim(spec: torch.Tensor, angles: np.ndarray, n_fft, win_length, hop_length, n_iter) -> torch.Tensor:
    """spec [F, T] magnitudes; ``angles`` = the reference's initial random phases
    np.angle(np.exp(2j*pi*np.random.rand(F, T))) (vocoder.py:101-102)."""
    s = spec.view(1, spec.shape[-2], spec.shape[-1])
    ang = torch.from_numpy(angles).to(s).view_as(s)
    wave = gl_inverse(s, ang, n_fft, win_length, hop_length).squeeze(1)
    for _ in range(n_iter):
        _, ang = gl_transform(wave, n_fft, win_length, hop_length)
        wave = gl_inverse(s, ang, n_fft, win_length, hop_length).squeeze(1)
    return wave.squeeze(0)


def initial_angles(shape, rs: np.random.RandomState) -> np.ndarray:
    return np.angle(np.exp(2j * np.pi * rs.rand(*shape))).astype(np.float32)


def slaney_mel_filters(sample_rate: int, n_fft: int, n_mels: int, f_min: float, f_max: float) -> torch.Tensor:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults htk=False, norm='slaney'
    (what audio_utils.py:234-242 calls) -- restated from librosa's documentation; un-pinned."""
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
        min_log_mel = min_log_hz / f_sp
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
        min_log_mel = min_log_hz / f_sp
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    if f_max is None:
        f_max = sample_rate / 2.0
    fft_f = np.linspace(0, sample_rate / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return torch.from_numpy(w.astype(np.float32))


def vocoder(feat: torch.Tensor, angles: np.ndarray, sample_rate, win_size, hop_size, n_fft, n_mels, f_min, f_max,
            n_iter) -> torch.Tensor:
    """GriffinLimVocoder.forward (vocoder.py:136-144): feat [T, n_mels] log-mel -> waveform."""
    x = feat.exp().transpose(-1, -2)
    basis = torch.pinverse(slaney_mel_filters(sample_rate, n_fft, n_mels, f_min, f_max))  # F x n_mels
    spec = basis.matmul(x).clamp(min=0)
    return griffin_lim(spec, angles, n_fft, win_size, hop_size, n_iter)


# ---- MCD (examples/s2s_trans/tasks/s2s_translation.py:414-552) ------------------------------------
def dtw(distance: torch.Tensor, shapes=None):
    """batch_dynamic_time_warping restated cell by cell (same first-minimum tie break over
    [left, up-left, up], same sequential cumsum initialisation, same backtrace)."""
    bsz, m, n = distance.shape
    cum = torch.zeros_like(distance)
    bp = torch.zeros(bsz, m, n, dtype=torch.int32) - 1
    cum[:, 0, :] = distance[:, 0, :].cumsum(-1)
    cum[:, :, 0] = distance[:, :, 0].cumsum(-1)
    bp[:, 0, :] = 0
    bp[:, :, 0] = 2
    for i in range(1, m):
        for j in range(1, n):
            c = torch.stack([cum[:, i, j - 1], cum[:, i - 1, j - 1], cum[:, i - 1, j]], dim=1)
            v, k = c.min(dim=1)
            bp[:, i, j] = k.int()
            cum[:, i, j] = v + distance[:, i, j]
    pm = torch.zeros_like(bp)
    for b in range(bsz):
        i = m - 1 if shapes is None else int(shapes[b][0]) - 1
        j = n - 1 if shapes is None else int(shapes[b][1]) - 1
        pm[b, i, j] = 1
        steps = 1
        while (i != 0 or j != 0) and steps < 10000:
            di, dj = {0: (0, -1), 1: (-1, -1), 2: (-1, 0)}[int(bp[b, i, j])]
            i, j = i + di, j + dj
            pm[b, i, j] = 1
            steps += 1
    return cum, bp, pm


def mfcc(y: torch.Tensor, sr: int, n_mfcc=13, n_mels=80, f_min=20.0) -> torch.Tensor:
    """torchaudio.transforms.MFCC(sr, n_mfcc, log_mels=True, melkwargs=...) restated from its documented
    defaults (un-pinned: torchaudio is absent) -> [T, n_mfcc]."""
    n_fft = win = int(0.05 * sr)
    hop = int(0.0125 * sr)
    spec = torch.stft(y.double(), n_fft, hop, win, window=torch.hann_window(win, periodic=True, dtype=torch.float64),
                      center=True, pad_mode="reflect", return_complex=True).abs() ** 2  # [F, T]
    F_ = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sr // 2, F_, dtype=torch.float64)
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    m_pts = torch.linspace(hz2mel(f_min), hz2mel(sr / 2.0), n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    fb = torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)
    mel = torch.log(spec.t() @ fb + 1e-6)
    k = torch.arange(n_mfcc, dtype=torch.float64).unsqueeze(1)
    dct = torch.cos(math.pi / n_mels * (torch.arange(n_mels, dtype=torch.float64) + 0.5) * k)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / n_mels)
    return (mel @ dct.t()).float()


def mcd(y1, y2, sr):
    x1, x2 = mfcc(y1, sr), mfcc(y2, sr)
    d = (torch.cdist(x1.unsqueeze(0), x2.unsqueeze(0), p=2).squeeze(0).pow(2) / x1.size(1)).pow(0.5)
    cum, bp, pm = dtw(d.unsqueeze(0))
    return float(cum[0, -1, -1] / pm[0].sum())
