"""`torchaudio.transforms.MFCC(sr, n_mfcc=13, log_mels=True, melkwargs=...)` as
examples/s2s_trans/tasks/s2s_translation.py:543-545 builds it, answered by oracle/infer_oracle.py: mfcc (a restatement of
torchaudio's documented defaults: power spectrogram, HTK mel filterbank without normalisation, log(mel + 1e-6),
orthonormal DCT-II) -- the transform itself is NOT pinned by this; the reference's DTW / distance / normalisation around it is."""
import torch

import infer_oracle as _IO


class MFCC(torch.nn.Module):
    def __init__(self, sample_rate=16000, n_mfcc=40, dct_type=2, norm="ortho", log_mels=False, melkwargs=None):
        super().__init__()
        mk = dict(melkwargs or {})
        want = {"n_fft": int(0.05 * sample_rate), "win_length": int(0.05 * sample_rate),
                "hop_length": int(0.0125 * sample_rate)}
        # the restatement covers the one configuration the reference asks for
        assert log_mels and dct_type == 2 and norm == "ortho", "only the reference's MFCC configuration is restated"
        for k, v in want.items():
            assert mk.get(k) == v, (k, mk.get(k), v)
        assert mk.get("window_fn", torch.hann_window) is torch.hann_window
        self.sample_rate, self.n_mfcc = sample_rate, n_mfcc
        self.n_mels, self.f_min = int(mk.get("n_mels", 128)), float(mk.get("f_min", 0.0))

    def forward(self, y):  # [N] -> [n_mfcc, T] (torchaudio's layout; the reference transposes it)
        return _IO.mfcc(y, self.sample_rate, self.n_mfcc, self.n_mels, self.f_min).transpose(-1, -2)
