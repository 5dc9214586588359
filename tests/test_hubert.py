"""Frozen HuBERT front end: the HIP path (through the C ABI,
s2st_hubert_*) against the oracle and against golden vectors produced by the reference's own
HubertModel (oracle/gen_golden_hubert.py)."""
import importlib
import os

import numpy as np
import pytest
import torch

import hubert_oracle as HO

HUB = "speech-to-speech-translation_amd.models.hubert"


def make_frontend(backend, cfg, precise):
    M = importlib.import_module(HUB)
    f = M.HubertFrontend(backend.device, conv=cfg["conv"], embed=cfg["embed"], layers=cfg["layers"], heads=cfg["heads"],
                         ffn=cfg["ffn"], conv_pos=cfg["conv_pos"], conv_pos_groups=cfg["conv_pos_groups"], precise=precise)
    f.load_state_dict(HO.synth_state(cfg))
    return f


def test_oracle_against_reference_golden(golden_dir):
    """CPU: the restatement reproduces what the reference HubertModel produced (pins the oracle)."""
    for name in ("tiny", "base"):
        z = np.load(os.path.join(golden_dir, f"hubert_{name}.npz"))
        cfg = HO.HUBERT_CONFIGS[name]
        wave, pad, _ = HO.synth_audio(int(z["B"]), int(z["N"]), int(z["seed"]))
        y, fpm = HO.extract_features(HO.synth_state(cfg), cfg, wave, pad)
        assert np.array_equal(fpm.numpy(), z["frame_pad"])
        if name == "tiny":
            np.testing.assert_allclose(y.numpy(), z["out"], rtol=0, atol=2e-5 * np.abs(z["out"]).max())
        else:
            np.testing.assert_allclose(y.numpy()[:, ::5, ::16], z["sample"], rtol=0, atol=2e-5 * np.abs(z["sample"]).max())
            assert abs(float(y.abs().sum()) - z["sum"][1]) < 1e-4 * z["sum"][1]


@pytest.mark.parametrize("precise", [True, False], ids=["bf16x3", "bf16"])
def test_tiny_hubert_vs_golden_and_oracle(backend, golden_dir, precise):
    """Whole front end on the tiny geometry: conv0 + GroupNorm + GELU, 6 conv GEMMs, LN, projection,
    padded-frame zeroing, grouped weight-normed pos-conv (+SamePad), 2 post-LN layers with key
    padding.  Tolerances: bf16x3 GEMMs 2e-4 of the output scale; bf16 operands 3e-2."""
    z = np.load(os.path.join(golden_dir, "hubert_tiny.npz"))
    cfg = HO.TINY
    f = make_frontend(backend, cfg, precise)
    wave, pad, lens = HO.synth_audio(int(z["B"]), int(z["N"]), int(z["seed"]))
    y, fpm = f.extract_features(wave, pad)
    backend.sync()
    assert np.array_equal(fpm.cpu().numpy(), z["frame_pad"])  # bit-exact (integer / bool work)
    ref = torch.from_numpy(z["out"])
    tol = 2e-4 if precise else 3e-2
    valid = ~torch.from_numpy(z["frame_pad"])
    err = ((y.cpu() - ref).abs() * valid.unsqueeze(-1)).max() / ref.abs().max()
    assert float(err) < tol, float(err)
    # state_dict round trip in the reference's names / layouts
    sd = f.state_dict()
    for k, v in HO.synth_state(cfg).items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        assert float((sd[k].cpu() - v).abs().max()) == 0.0, k


def test_tiny_hubert_other_batch_against_oracle(backend):
    """A second seeded batch (different lengths, no golden): HIP path vs the oracle."""
    cfg = HO.TINY
    f = make_frontend(backend, cfg, True)
    wave, pad, _ = HO.synth_audio(2, 2500, 21)
    y, fpm = f.extract_features(wave, pad)
    backend.sync()
    yo, fo = HO.extract_features(HO.synth_state(cfg), cfg, wave, pad)
    assert torch.equal(fpm.cpu(), fo)
    err = ((y.cpu() - yo).abs() * (~fo).unsqueeze(-1)).max() / yo.abs().max()
    assert float(err) < 2e-4, float(err)


@pytest.mark.gpu
@pytest.mark.parametrize("precise", [True, False], ids=["bf16x3", "bf16"])
def test_base_hubert_golden(backend, golden_dir, precise):
    """hubert_base geometry (7 convs x 512, 12 x 768, pos-conv k128 g16) against the reference's golden."""
    if backend.kind != "hip":
        pytest.skip("base geometry runs on the GPU")
    z = np.load(os.path.join(golden_dir, "hubert_base.npz"))
    cfg = HO.BASE
    f = make_frontend(backend, cfg, precise)
    wave, pad, _ = HO.synth_audio(int(z["B"]), int(z["N"]), int(z["seed"]))
    y, fpm = f.extract_features(wave, pad)
    backend.sync()
    assert np.array_equal(fpm.cpu().numpy(), z["frame_pad"])
    valid = (~torch.from_numpy(z["frame_pad"]))[:, ::5].unsqueeze(-1)
    ref = torch.from_numpy(z["sample"])
    err = ((y.cpu()[:, ::5, ::16] - ref).abs() * valid).max() / ref.abs().max()
    assert float(err) < (3e-4 if precise else 4e-2), float(err)


def test_frame_padding_mask_fast_path_equals_the_reduction():
    """hubert.py:400-410 (a frame is padding iff all samples of its chunk are) for suffix masks from the row's valid-sample
    count (HubertFrontend._suffix_frame_mask, the staging path of every --use-hubert batch) == the [B][T][chunk] reduction
    the reference does, over random lengths / frame counts incl. rows without padding, fully padded rows and lengths that
    are not a multiple of the frame count; a mask whose padding is not a suffix is left to the general path."""
    F = importlib.import_module("speech-to-speech-translation_amd.models.hubert").HubertFrontend
    rs = np.random.RandomState(0)
    for trial in range(300):
        B, N = rs.randint(1, 5), rs.randint(50, 4000)
        T = rs.randint(1, min(N, 60))
        pm = torch.zeros(B, N, dtype=torch.bool)
        for b in range(B):
            k = N if rs.rand() < 0.2 else rs.randint(0, N + 1)
            pm[b, k:] = True
        fast = F._suffix_frame_mask(pm, T)
        assert fast is not None and torch.equal(F.frame_padding_mask(pm, T), fast), (trial, B, N, T)
    pm = torch.zeros(2, 100, dtype=torch.bool)
    pm[0, 10:20] = True
    assert F._suffix_frame_mask(pm, 7) is None


@pytest.mark.parametrize("k,stride,C,T", [(10, 5, 32, 300), (7, 3, 8, 130), (3, 1, 4, 1), (16, 8, 64, 129)])
def test_first_block_kernels(backend, k, stride, C, T):
    """Conv1d(1, C, k, stride, bias=False) -> GroupNorm(C, C) -> GELU (wav2vec2.py:777-783, 806-814) through the C ABI against
    float64 numpy: the 10-tap form HuBERT uses and the general one (zero-weight taps, clamped reads), several time blocks of
    128 frames, a single frame (variance 0), a waveform with a large offset (the one-pass statistics work on deviations
    from the channel's first frame), fp32 and bf16 outputs."""
    import ctypes as C_
    from math import erf
    rs = np.random.RandomState(k * 100 + stride)
    B = 3
    N = (T - 1) * stride + k
    wave = (rs.randn(B, N) + np.array([0.0, 40.0, -3.0])[:, None]).astype(np.float32)
    w = (rs.randn(C, k) / np.sqrt(k)).astype(np.float32)
    g = (1.0 + 0.2 * rs.randn(C)).astype(np.float32)
    b_ = (0.1 * rs.randn(C)).astype(np.float32)
    idx = np.arange(T)[:, None] * stride + np.arange(k)[None, :]
    conv = np.einsum("btk,ck->btc", wave.astype(np.float64)[:, idx], w.astype(np.float64))
    mu, var = conv.mean(1, keepdims=True), conv.var(1, keepdims=True)
    z = (conv - mu) / np.sqrt(var + 1e-5) * g + b_
    ref = 0.5 * z * (1.0 + np.vectorize(erf)(z / np.sqrt(2.0)))
    dev = backend.device
    lib = backend.bd.lib()
    lib.s2st_hubert_conv0_stats_floats_i64.restype = C_.c_int64
    lib.s2st_hubert_conv0_stats_floats_i64.argtypes = [C_.c_int32] * 3
    stats = torch.empty(int(lib.s2st_hubert_conv0_stats_floats_i64(B, T, C)), device=dev)
    y = torch.empty(B, T, C, device=dev)
    yh = torch.empty(B, T, C, dtype=torch.bfloat16, device=dev)
    backend.bd.call("s2st_hubert_conv0_gn_gelu_f32", torch.from_numpy(wave).to(dev), torch.from_numpy(w).to(dev),
                    torch.from_numpy(g).to(dev), torch.from_numpy(b_).to(dev), y, yh, stats, B, N, T, C, k, stride, 1e-5)
    backend.sync()
    # (an utterance with offset 40: conv values ~ 40 * sum(w) with unit spread; the statistics must not lose the spread)
    assert float(np.abs(y.cpu().numpy() - ref).max()) < 2e-4 * max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(yh.float().cpu().numpy() - ref).max()) < 1e-2 * max(1.0, float(np.abs(ref).max()))
