"""`librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` as fairseq/data/audio/audio_utils.py:241 calls it (positionally;
librosa's defaults htk=False, norm='slaney'), answered by oracle/infer_oracle.py: slaney_mel_filters -- the table itself is
NOT pinned by this (librosa is absent); what the stand-in makes runnable is the reference code around it."""
import infer_oracle as _IO


def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    return _IO.slaney_mel_filters(sr, n_fft, n_mels, fmin, fmax).numpy()
