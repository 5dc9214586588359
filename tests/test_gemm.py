"""s2st_gemm_f32 against torch fp32 matmul: every operand layout, both precisions,
ragged sizes, split rows (halo buffers), batching, fused epilogues."""
import numpy as np
import pytest
import torch


def _ref(A, B):
    return A.double() @ B.double().t()


def _relerr(C, R):
    return ((C.double().cpu() - R).abs().max() / R.abs().max()).item()


SHAPES = [(100, 72, 96), (130, 74, 44), (67, 130, 50), (257, 192, 320)]


@pytest.mark.parametrize("akm", [True, False])
@pytest.mark.parametrize("bkm", [True, False])
@pytest.mark.parametrize("precise", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_layouts(backend, akm, bkm, precise, shape):
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    Am = (A if akm else A.t().contiguous()).to(backend.device)
    Bm = (B if bkm else B.t().contiguous()).to(backend.device)
    C = torch.full((M, N), 7.0, device=backend.device)
    backend.bd.gemm(Am, Bm, C, M, N, K, a_kmajor=akm, b_kmajor=bkm, precise=precise)
    backend.sync()
    # tolerance: bf16 inputs (8-bit mantissa) vs bf16x3 split (~fp32)
    assert _relerr(C, _ref(A, B)) < (2e-5 if precise else 1.5e-2)


def test_big_tile_and_splitk(backend):
    if backend.kind == "emu":
        M, N, K = 256, 256, 64
    else:
        M, N, K = 4096, 1024, 512
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    C = torch.zeros(M, N, device=backend.device)
    backend.bd.gemm(A.to(backend.device), B.to(backend.device), C, M, N, K)
    backend.sync()
    assert _relerr(C, _ref(A, B)) < 1.5e-2
    # weight-gradient form: dW[N,K] += dY[M,N]^T X[M,K], reduction over M, accumulate (split-K)
    Mred = 1024 if backend.kind == "emu" else 8192
    dY = torch.randn(Mred, 64, generator=g)
    X = torch.randn(Mred, 96, generator=g)
    dW0 = torch.randn(64, 96, generator=g)
    dW = dW0.clone().to(backend.device)
    backend.bd.gemm(dY.to(backend.device), X.to(backend.device), dW, 64, 96, Mred,
                    a_kmajor=False, a_ld=64, b_kmajor=False, b_ld=96, accumulate=True, precise=True)
    backend.sync()
    R = dW0.double() + dY.double().t() @ X.double()
    assert _relerr(dW, R) < 3e-5


def test_epilogue_bias_relu_resid_alpha(backend):
    M, N, K = 90, 70, 64
    g = torch.Generator().manual_seed(2)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    d = backend.device
    C = torch.zeros(M, N, device=d)
    backend.bd.gemm(A.to(d), B.to(d), C, M, N, K, alpha=0.5, bias=bias.to(d), act=1,
                    resid=res.to(d), precise=True)
    backend.sync()
    R = torch.relu(0.5 * _ref(A, B) + bias.double()) + res.double()
    assert _relerr(C, R) < 2e-5
    # accumulate without split
    C2 = res.clone().to(d)
    backend.bd.gemm(A.to(d), B.to(d), C2, M, N, K, accumulate=True, precise=True)
    backend.sync()
    assert _relerr(C2, res.double() + _ref(A, B)) < 2e-5


def test_conv_as_gemm_halo(backend):
    """Conv1d(k=5, stride s, pad 2) == GEMM over a halo-padded [B][T+4][C] buffer whose
    im2col rows are overlapping windows (s2st_transformer.py:135-139)."""
    Bn, T, Cin, Cout, Kw = 3, 37, 8, 24, 5
    g = torch.Generator().manual_seed(3)
    x = torch.randn(Bn, T, Cin, generator=g)
    w = torch.randn(Cout, Cin, Kw, generator=g)
    b = torch.randn(Cout, generator=g)
    d = backend.device
    for stride in (1, 2):
        ref = torch.nn.functional.conv1d(x.transpose(1, 2), w, b, stride=stride, padding=2).transpose(1, 2)
        Tout = ref.shape[1]
        xp = torch.zeros(Bn, T + 4, Cin)
        xp[:, 2:T + 2] = x
        wf = w.permute(0, 2, 1).contiguous()  # [O][Kw][I]
        y = torch.zeros(Bn, Tout, Cout, device=d)
        backend.bd.gemm(xp.to(d), wf.to(d), y, Bn * Tout, Cout, Kw * Cin,
                        a_ld=stride * Cin, a_per=Tout, a_bs=(T + 4) * Cin, b_ld=Kw * Cin,
                        bias=b.to(d), precise=True)
        backend.sync()
        assert _relerr(y, ref.double()) < 2e-5


def test_batched_attention_forms(backend):
    """Q K^T and P V with (b, h) batching over [B*T, C] projections
    (multihead_attention.py:332, 367)."""
    Bn, H, T, S, Dh = 2, 4, 19, 23, 16
    Cm = H * Dh
    g = torch.Generator().manual_seed(4)
    q = torch.randn(Bn, T, Cm, generator=g)
    k = torch.randn(Bn, S, Cm, generator=g)
    v = torch.randn(Bn, S, Cm, generator=g)
    d = backend.device
    ld = 24  # padded score rows
    sc = torch.zeros(Bn, H, T, ld, device=d)
    backend.bd.gemm(q.to(d), k.to(d), sc, T, S, Dh, a_ld=Cm, a_zo=T * Cm, a_zi=Dh,
                    b_ld=Cm, b_zo=S * Cm, b_zi=Dh, c_ld=ld, c_zo=H * T * ld, c_zi=T * ld,
                    batch=Bn * H, zdiv=H, alpha=0.25, precise=True)
    backend.sync()
    qh = q.view(Bn, T, H, Dh).permute(0, 2, 1, 3).double()
    kh = k.view(Bn, S, H, Dh).permute(0, 2, 1, 3).double()
    vh = v.view(Bn, S, H, Dh).permute(0, 2, 1, 3).double()
    R = 0.25 * qh @ kh.transpose(-1, -2)
    assert _relerr(sc[..., :S], R) < 2e-5
    p = torch.softmax(R, -1).float()
    pp = torch.zeros(Bn, H, T, ld)
    pp[..., :S] = p
    o = torch.zeros(Bn, T, Cm, device=d)
    backend.bd.gemm(pp.to(d), v.to(d), o, T, Dh, S, a_ld=ld, a_zo=H * T * ld, a_zi=T * ld,
                    b_kmajor=False, b_ld=Cm, b_zo=S * Cm, b_zi=Dh,
                    c_ld=Cm, c_zo=T * Cm, c_zi=Dh, batch=Bn * H, zdiv=H, precise=True)
    backend.sync()
    Ro = (p.double() @ vh).permute(0, 2, 1, 3).reshape(Bn, T, Cm)
    assert _relerr(o, Ro) < 2e-5


def test_dropout_epilogue_statistics(backend):
    M, N, K = 128, 128, 32
    d = backend.device
    A = torch.ones(M, K, device=d)
    B = torch.ones(N, K, device=d) / K
    C = torch.zeros(M, N, device=d)
    backend.bd.gemm(A, B, C, M, N, K, drop_p=0.25, seed=1234, precise=True)
    backend.sync()
    C = C.cpu()
    kept = (C != 0)
    assert abs(kept.float().mean().item() - 0.75) < 0.02
    np.testing.assert_allclose(C[kept].numpy(), 1.0 / 0.75, rtol=1e-5)
    C2 = torch.zeros(M, N, device=d)
    backend.bd.gemm(A, B, C2, M, N, K, drop_p=0.25, seed=1234, precise=True)
    backend.sync()
    assert torch.equal(C, C2.cpu())  # same seed -> same mask


# ---- bf16-operand fast path (gemm_bf16.hip) ---------------------------------------------------
def _bf(x):
    return x.to(torch.bfloat16)


def _pad_cols(x, mult=8):
    """[R][C] -> [R][round_up(C, mult)] with NaN padding: the kernel must ignore pad content."""
    R, Cc = x.shape
    ld = (Cc + mult - 1) // mult * mult
    out = torch.full((R, ld), float("nan"), dtype=x.dtype)
    out[:, :Cc] = x
    return out, ld


BF_SHAPES = [(100, 72, 96), (130, 74, 44), (67, 130, 50), (257, 192, 320), (191, 128, 191)]


@pytest.mark.parametrize("akm", [True, False])
@pytest.mark.parametrize("bkm", [True, False])
@pytest.mark.parametrize("shape", BF_SHAPES)
def test_bf16_layouts(backend, akm, bkm, shape):
    """bf16 operands, every layout, ragged M/N/K; padded lds filled with NaN (contract: pad
    content is never used); fp32 result + bf16 copy."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + 1)
    A = _bf(torch.randn(M, K, generator=g))
    B = _bf(torch.randn(N, K, generator=g))
    Am, a_ld = _pad_cols(A if akm else A.t().contiguous())
    Bm, b_ld = _pad_cols(B if bkm else B.t().contiguous())
    d = backend.device
    ldc = (N + 3) // 4 * 4
    C = torch.full((M, ldc), 7.0, device=d)
    Ch = torch.zeros(M, ldc, dtype=torch.bfloat16, device=d)
    backend.bd.gemm(Am.to(d), Bm.to(d), C, M, N, K, a_kmajor=akm, b_kmajor=bkm, a_ld=a_ld, b_ld=b_ld,
                    c_ld=ldc, c_bf16=Ch)
    backend.sync()
    R = A.double() @ B.double().t()
    assert _relerr(C[:, :N], R) < 2e-6  # exact products of bf16 values, fp32 accumulation
    assert torch.equal(Ch[:, :N].cpu(), C[:, :N].cpu().to(torch.bfloat16))


def test_bf16_matches_f32_path_bitwise_inputs(backend):
    """Same rounding as the fp32-operand kernel: feeding bf16(x) gives what feeding x gives."""
    M, N, K = 96, 80, 128
    g = torch.Generator().manual_seed(11)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    d = backend.device
    C1 = torch.zeros(M, N, device=d)
    C2 = torch.zeros(M, N, device=d)
    backend.bd.gemm(A.to(d), B.to(d), C1, M, N, K)
    backend.bd.gemm(_bf(A).to(d), _bf(B).to(d), C2, M, N, K)
    backend.sync()
    assert _relerr(C2, C1.double().cpu()) < 1e-6


def test_bf16_epilogue_and_unaligned(backend):
    M, N, K = 90, 70, 64
    g = torch.Generator().manual_seed(12)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g))
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    d = backend.device
    C = torch.zeros(M, N, device=d)  # ld 70: scalar epilogue path
    backend.bd.gemm(A.to(d), B.to(d), C, M, N, K, alpha=0.5, bias=bias.to(d), act=1, resid=res.to(d))
    backend.sync()
    R = torch.relu(0.5 * (A.double() @ B.double().t()) + bias.double()) + res.double()
    assert _relerr(C, R) < 2e-6
    C2 = res.clone().to(d)
    backend.bd.gemm(A.to(d), B.to(d), C2, M, N, K, accumulate=True)
    backend.sync()
    assert _relerr(C2, res.double() + A.double() @ B.double().t()) < 2e-6
    # unaligned ld (K = 50 -> ld 50): guarded scalar loader
    A3, B3 = _bf(torch.randn(33, 50, generator=g)), _bf(torch.randn(21, 50, generator=g))
    C3 = torch.zeros(33, 21, device=d)
    backend.bd.gemm(A3.to(d), B3.to(d), C3, 33, 21, 50)
    backend.sync()
    assert _relerr(C3, A3.double() @ B3.double().t()) < 2e-6


def test_bf16_tiles_splitk_conv_batched(backend):
    d = backend.device
    g = torch.Generator().manual_seed(13)
    # large-tile selection (128x128 / 128x64) and the wgrad split-K form
    M, N, K = (256, 256, 128) if backend.kind == "emu" else (4584, 512, 2048)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g))
    C = torch.zeros(M, N, device=d)
    backend.bd.gemm(A.to(d), B.to(d), C, M, N, K)
    backend.sync()
    assert _relerr(C, A.double() @ B.double().t()) < 2e-6
    Mred = 1024 if backend.kind == "emu" else 8192
    dY, X = _bf(torch.randn(Mred, 64, generator=g)), _bf(torch.randn(Mred, 96, generator=g))
    dW0 = torch.randn(64, 96, generator=g)
    dW = dW0.clone().to(d)
    backend.bd.gemm(dY.to(d), X.to(d), dW, 64, 96, Mred, a_kmajor=False, a_ld=64, b_kmajor=False, b_ld=96,
                    accumulate=True)
    backend.sync()
    assert _relerr(dW, dW0.double() + dY.double().t() @ X.double()) < 1e-5
    # same with caller scratch: partial slabs + combine kernel instead of atomics
    dW2 = dW0.clone().to(d)
    ws = torch.empty(1 << 20, device=d)
    backend.bd.gemm(dY.to(d), X.to(d), dW2, 64, 96, Mred, a_kmajor=False, a_ld=64, b_kmajor=False, b_ld=96,
                    accumulate=True, ws=ws)
    backend.sync()
    assert _relerr(dW2, dW0.double() + dY.double().t() @ X.double()) < 1e-5
    # conv as GEMM over a halo image (stride 2), bf16
    Bn, T, Cin, Cout, Kw = 3, 37, 8, 24, 5
    x = _bf(torch.randn(Bn, T, Cin, generator=g))
    w = _bf(torch.randn(Cout, Cin, Kw, generator=g))
    ref = torch.nn.functional.conv1d(x.float().transpose(1, 2), w.float(), None, stride=2, padding=2).transpose(1, 2)
    Tout = ref.shape[1]
    xp = torch.zeros(Bn, T + 4, Cin, dtype=torch.bfloat16)
    xp[:, 2:T + 2] = x
    wf = w.permute(0, 2, 1).contiguous()
    y = torch.zeros(Bn, Tout, Cout, device=d)
    backend.bd.gemm(xp.to(d), wf.to(d), y, Bn * Tout, Cout, Kw * Cin, a_ld=2 * Cin, a_per=Tout,
                    a_bs=(T + 4) * Cin, b_ld=Kw * Cin)
    backend.sync()
    assert _relerr(y, ref.double()) < 2e-6
    # batched P V with rows-contiguous V (b, h batching), S not a multiple of 8
    Bn, H, T, S, Dh = 2, 4, 19, 23, 16
    Cm = H * Dh
    p = _bf(torch.rand(Bn, H, T, S, generator=g))
    v = _bf(torch.randn(Bn, S, Cm, generator=g))
    ld = 24
    pp = torch.full((Bn, H, T, ld), float("nan"), dtype=torch.bfloat16)
    pp[..., :S] = p
    o = torch.zeros(Bn, T, Cm, device=d)
    backend.bd.gemm(pp.to(d), v.to(d), o, T, Dh, S, a_ld=ld, a_zo=H * T * ld, a_zi=T * ld, b_kmajor=False,
                    b_ld=Cm, b_zo=S * Cm, b_zi=Dh, c_ld=Cm, c_zo=T * Cm, c_zi=Dh, batch=Bn * H, zdiv=H)
    backend.sync()
    vh = v.view(Bn, S, H, Dh).permute(0, 2, 1, 3).double()
    Ro = (p.double() @ vh).permute(0, 2, 1, 3).reshape(Bn, T, Cm)
    assert _relerr(o, Ro) < 2e-6


@pytest.mark.parametrize("M,N,K,act,resid", [(16, 512, 512, 0, False), (16, 2048, 512, 1, False), (16, 512, 2048, 0, True),
                                             (5, 80, 256, 0, False), (1, 1, 32, 2, True), (16, 1536, 512, 0, False),
                                             # round 4: up to 64 rows (2 / 4 row blocks per lane), logistic epilogue
                                             (17, 512, 512, 0, True), (32, 1536, 512, 1, False), (49, 80, 256, 0, False),
                                             (64, 512, 2048, 0, True), (64, 1, 512, 3, False), (16, 1, 512, 3, False)])
def test_skinny_gemm(backend, M, N, K, act, resid):
    """AR-decoding product (s2st_gemm_skinny_f32): bf16-rounded operands, fp32 accumulation, bias / activation /
    residual -- against the same product formed in double from the rounded operands."""
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M + 2, K + 8, generator=g)[:M, :K]  # non-contiguous rows: exercises the row strides
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if resid else None
    dev = backend.device
    xd = torch.zeros(M + 2, K + 8, device=dev)
    xd[:M, :K] = x.to(dev)
    wd, bd_, y = w.to(dev), b.to(dev), torch.full((M, N + 4), float("nan"), device=dev)
    rd = r.to(dev) if resid else None
    backend.bd.call("s2st_gemm_skinny_f32", xd, K + 8, wd, K, y, N + 4, bd_, act, 0.0, 0, rd, N, M, N, K)
    backend.sync()
    ref = x.to(torch.bfloat16).double() @ w.double().t() + b.double()
    if act == 1:
        ref = ref.clamp_min(0)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    elif act == 3:
        ref = torch.sigmoid(ref)
    if resid:
        ref = ref + r.double()
    got = y[:, :N].cpu().double()
    assert torch.isnan(y[:, N:]).all()
    assert float((got - ref).abs().max()) <= 2e-5 * (float(ref.abs().max()) + 1.0)
    # dropout: same mask indexing as the tiled kernels' epilogue (element index m * N + n)
    if act == 0 and not resid and N % 8 == 0:
        y2 = torch.zeros(M, N, device=dev)
        backend.bd.call("s2st_gemm_skinny_f32", xd, K + 8, wd, K, y2, N, bd_, 0, 0.5, 77, None, N, M, N, K)
        y3 = torch.zeros(M, N, device=dev)
        backend.bd.gemm(xd[:M].contiguous()[:, :K].contiguous().to(torch.bfloat16), wd, y3, M, N, K, bias=bd_, drop_p=0.5, seed=77)
        backend.sync()
        assert torch.equal(y2 == 0, y3 == 0) and float((y2 - y3).abs().max()) <= 1e-4 * (float(y3.abs().max()) + 1.0)


@pytest.mark.parametrize("M,N,K,act", [(16, 1536, 512, 0), (16, 2048, 512, 1), (3, 80, 256, 0), (16, 1, 512, 0),
                                       (64, 1536, 512, 0), (40, 320, 512, 1), (64, 1, 512, 3), (23, 2048, 512, 1)])
def test_skinny_gemm_with_fused_layernorm(backend, M, N, K, act):
    """s2st_ln_gemm_skinny_f32 == LayerNorm (fp32) -> bf16 rounding -> product, formed in double."""
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(M, K, generator=g) * 2 + 0.5
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b, gam, bet = torch.randn(N, generator=g), torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    dev = backend.device
    y = torch.zeros(M, N, device=dev)
    backend.bd.call("s2st_ln_gemm_skinny_f32", x.to(dev), K, gam.to(dev), bet.to(dev), 1e-5, w.to(dev), K, y, N, b.to(dev), act, M, N, K)
    backend.sync()
    ln = torch.nn.functional.layer_norm(x, (K,), gam, bet, 1e-5)
    ref = ln.to(torch.bfloat16).double() @ w.double().t() + b.double()
    if act == 1:
        ref = ref.clamp_min(0)
    elif act == 3:
        ref = torch.sigmoid(ref)
    # a last-bit difference in the statistics can flip the bf16 rounding of single inputs: 2^-9 of one product term
    assert float((y.cpu().double() - ref).abs().max()) <= 2e-3 * (float(ref.abs().max()) + 1.0)
    assert float((y.cpu().double() - ref).abs().mean()) <= 1e-4 * (float(ref.abs().max()) + 1.0)


def _needs_experimental(backend):
    """The persistent tile walk / stream-K / 256 x 128 forms are compiled only into -DS2ST_EXPERIMENTAL builds (the
    emulator's test build, tools/build_experimental.sh); the product library ignores their switches."""
    fn = backend.bd.lib().s2st_experimental_build
    fn.restype = __import__("ctypes").c_int
    if not fn():
        pytest.skip("forms of -DS2ST_EXPERIMENTAL builds only")


@pytest.mark.parametrize("akm,bkm", [(True, True), (True, False), (False, False), (False, True)])
@pytest.mark.parametrize("K", [64, 200, 512])
def test_bf16_persistent_kernel(backend, monkeypatch, akm, bkm, K):
    """The persistent ring kernel (more tiles than workgroups: every workgroup walks several tiles with the DMA ring
    running across tile boundaries -- K = 64 makes the prologue itself span tiles, K = 200 has a K tail) against the
    exact product; epilogue variants with both output copies."""
    _needs_experimental(backend)
    monkeypatch.setenv("S2ST_GEMM_PERSIST", "2")
    monkeypatch.setenv("S2ST_GEMM_TILE", "128x128")  # (the small emulator shape would get 64 x 64 tiles: one-shot kernel)
    M, N = (640, 256) if backend.kind == "emu" else (4584, 2048)
    g = torch.Generator().manual_seed(K + 2 * akm + bkm)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g))
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Am, a_ld = _pad_cols(A if akm else A.t().contiguous())
    Bm, b_ld = _pad_cols(B if bkm else B.t().contiguous())
    d = backend.device
    R = A.double() @ B.double().t()
    C = torch.full((M, N), 7.0, device=d)
    Ch = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
    backend.bd.gemm(Am.to(d), Bm.to(d), C, M, N, K, a_kmajor=akm, b_kmajor=bkm, a_ld=a_ld, b_ld=b_ld, c_bf16=Ch)
    backend.sync()
    assert _relerr(C, R) < 2e-6
    assert torch.equal(Ch.cpu(), C.cpu().to(torch.bfloat16))
    C2 = torch.zeros(M, N, device=d)
    backend.bd.gemm(Am.to(d), Bm.to(d), C2, M, N, K, a_kmajor=akm, b_kmajor=bkm, a_ld=a_ld, b_ld=b_ld, alpha=0.5,
                    bias=bias.to(d), act=1, resid=res.to(d))
    backend.sync()
    assert _relerr(C2, torch.relu(0.5 * R + bias.double()) + res.double()) < 2e-6
    # the one-shot kernel gives the same bits (same products, same summation order)
    monkeypatch.setenv("S2ST_GEMM_PERSIST", "0")
    C3 = torch.zeros(M, N, device=d)
    backend.bd.gemm(Am.to(d), Bm.to(d), C3, M, N, K, a_kmajor=akm, b_kmajor=bkm, a_ld=a_ld, b_ld=b_ld)
    backend.sync()
    assert torch.equal(C3, C)


@pytest.mark.parametrize("akm,bkm", [(True, True), (True, False), (False, False), (False, True)])
@pytest.mark.parametrize("K", [64, 200, 512])
@pytest.mark.parametrize("tile", ["128x128", "128x64"])
def test_bf16_w4_early_release_kernel(backend, monkeypatch, akm, bkm, K, tile):
    """The 4-wave early-release ring form (gemm_bf16_w4.hip; S2ST_GEMM_W4=1): K = 64 is a single K-step (no refill), K =
    200 has a K tail behind refills, K = 512 runs the steady state; every operand layout, both tile shapes; epilogue
    variants with both output copies; bit-equal to the 8-wave ring kernel (same products, same summation order)."""
    monkeypatch.setenv("S2ST_GEMM_W4", "1")
    monkeypatch.setenv("S2ST_GEMM_PERSIST", "0")
    monkeypatch.setenv("S2ST_GEMM_TILE", tile)
    M, N = (384, 256) if backend.kind == "emu" else (4584, 1536)
    g = torch.Generator().manual_seed(K + 2 * akm + bkm)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g))
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Am, a_ld = _pad_cols(A if akm else A.t().contiguous())
    Bm, b_ld = _pad_cols(B if bkm else B.t().contiguous())
    d = backend.device
    R = A.double() @ B.double().t()
    C = torch.full((M, N), 7.0, device=d)
    Ch = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
    backend.bd.gemm(Am.to(d), Bm.to(d), C, M, N, K, a_kmajor=akm, b_kmajor=bkm, a_ld=a_ld, b_ld=b_ld, c_bf16=Ch)
    backend.sync()
    assert _relerr(C, R) < 2e-6
    assert torch.equal(Ch.cpu(), C.cpu().to(torch.bfloat16))
    C2 = torch.zeros(M, N, device=d)
    backend.bd.gemm(Am.to(d), Bm.to(d), C2, M, N, K, a_kmajor=akm, b_kmajor=bkm, a_ld=a_ld, b_ld=b_ld, alpha=0.5,
                    bias=bias.to(d), act=1, resid=res.to(d))
    backend.sync()
    assert _relerr(C2, torch.relu(0.5 * R + bias.double()) + res.double()) < 2e-6
    monkeypatch.setenv("S2ST_GEMM_W4", "0")
    C3 = torch.zeros(M, N, device=d)
    backend.bd.gemm(Am.to(d), Bm.to(d), C3, M, N, K, a_kmajor=akm, b_kmajor=bkm, a_ld=a_ld, b_ld=b_ld)
    backend.sync()
    assert torch.equal(C3, C)


@pytest.mark.parametrize("K", [64, 200, 512, 776])
def test_bf16_p4_four_phase_kernel(backend, monkeypatch, K):
    """The 256 x 256 four-phase form (gemm_bf16_p4.hip; forced by S2ST_GEMM_TILE=256x256): K = 64 is a single K-tile (no
    second buffer), K = 200 / 776 end in a K tail (an even and an odd number of K-tiles: both buffers hold the tail once), K =
    512 is the steady state; ragged M and N (partial tiles on both edges); epilogue variants with both output copies;
    bit-equal to the 8-wave ring kernel (same products, same summation order over K); falls back to the 128-row forms for
    operands it cannot take (a rows-contiguous B) and when switched off."""
    monkeypatch.setenv("S2ST_GEMM_PERSIST", "0")
    monkeypatch.setenv("S2ST_GEMM_TILE", "256x256")
    M, N = (300, 520) if backend.kind == "emu" else (4584, 1544)
    g = torch.Generator().manual_seed(K)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g))
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Am, a_ld = _pad_cols(A)
    Bm, b_ld = _pad_cols(B)
    d = backend.device
    R = A.double() @ B.double().t()
    C = torch.full((M, N), 7.0, device=d)
    Ch = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
    tile = backend.bd.gemm(Am.to(d), Bm.to(d), C, M, N, K, a_kmajor=True, b_kmajor=True, a_ld=a_ld, b_ld=b_ld, c_bf16=Ch,
                           return_tile=True)
    backend.sync()
    assert tile == (256, 256), tile
    assert _relerr(C, R) < 2e-6
    assert torch.equal(Ch.cpu(), C.cpu().to(torch.bfloat16))
    C2 = torch.zeros(M, N, device=d)
    backend.bd.gemm(Am.to(d), Bm.to(d), C2, M, N, K, a_kmajor=True, b_kmajor=True, a_ld=a_ld, b_ld=b_ld, alpha=0.5,
                    bias=bias.to(d), act=1, resid=res.to(d))
    backend.sync()
    assert _relerr(C2, torch.relu(0.5 * R + bias.double()) + res.double()) < 2e-6
    # bf16-only output with dropout (the FFN's hidden activation): the mask is the one the 128-row forms apply
    H1 = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
    backend.bd.gemm(Am.to(d), Bm.to(d), None, M, N, K, a_kmajor=True, b_kmajor=True, a_ld=a_ld, b_ld=b_ld, bias=bias.to(d), act=1,
                    drop_p=0.25, seed=77, c_bf16=H1)
    monkeypatch.setenv("S2ST_GEMM_P4", "0")
    H2 = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
    tile2 = backend.bd.gemm(Am.to(d), Bm.to(d), None, M, N, K, a_kmajor=True, b_kmajor=True, a_ld=a_ld, b_ld=b_ld, bias=bias.to(d),
                            act=1, drop_p=0.25, seed=77, c_bf16=H2, return_tile=True)
    C3 = torch.zeros(M, N, device=d)
    backend.bd.gemm(Am.to(d), Bm.to(d), C3, M, N, K, a_kmajor=True, b_kmajor=True, a_ld=a_ld, b_ld=b_ld)
    backend.sync()
    assert tile2 != (256, 256)
    assert torch.equal(H1, H2) and float((H1.float() == 0).float().mean()) > 0.3
    assert torch.equal(C3, C)
    monkeypatch.delenv("S2ST_GEMM_P4")
    # a rows-contiguous B: not this form's
    Bt = B.t().contiguous()
    Btm, bt_ld = _pad_cols(Bt)
    C4 = torch.zeros(M, N, device=d)
    tile4 = backend.bd.gemm(Am.to(d), Btm.to(d), C4, M, N, K, a_kmajor=True, b_kmajor=False, a_ld=a_ld, b_ld=bt_ld, return_tile=True)
    backend.sync()
    assert tile4 != (256, 256) and torch.equal(C4, C)


def test_bf16_p4_forced_on_a_split_k_product_falls_back(backend, monkeypatch):
    """with the four-phase form FORCED (S2ST_GEMM_P4=1 / S2ST_GEMM_TILE=256x256) an accumulating K-major x
    K-major product with fewer than 256 tiles splits K -- the 256 x 256 kernel has no split-K form, and the launcher used to
    fall through to a 64 x 64 launch on a grid built for 256 x 256 tiles (most of C never written).  It now goes back to
    128 x 128 tiles before the grid is derived: the accumulated result is the exact one, every element written."""
    M, N, K = (520, 300, 1100) if backend.kind == "emu" else (2560, 2560, 4608)
    g = torch.Generator().manual_seed(5)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g))
    Am, a_ld = _pad_cols(A)
    Bm, b_ld = _pad_cols(B)
    d = backend.device
    old = torch.randn(M, N, generator=g)
    ws = torch.zeros(16 << 20, device=d)
    ref = old.double() + A.double() @ B.double().t()
    for env in ({"S2ST_GEMM_TILE": "256x256"}, {"S2ST_GEMM_P4": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        C = old.clone().to(d)
        tile = backend.bd.gemm(Am.to(d), Bm.to(d), C, M, N, K, a_kmajor=True, b_kmajor=True, a_ld=a_ld, b_ld=b_ld, accumulate=True,
                               ws=ws, return_tile=True)
        backend.sync()
        assert tile != (256, 256), (env, tile)
        assert _relerr(C, ref) < 2e-6, env
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("tile", [128, 256, "oneshot", "w4"])
def test_bf16_group_of_weight_gradients(backend, monkeypatch, tile):
    """s2st_gemm_group_f32: a layer's weight-gradient products dW_i += dY_i^T X_i (different shapes, K = tokens, one with
    a K tail) in one launch, against the exact sums; 128 x 128 tiles and (S2ST_GROUP_TILE=256) 256 x 128 tiles with a
    ragged last tile row."""
    if tile in ("oneshot", "w4"):  # (default) one workgroup per tile of the concatenated list, plain K-loop
        monkeypatch.setenv("S2ST_GROUP_ONESHOT", "1")
        monkeypatch.setenv("S2ST_GEMM_W4", "2" if tile == "w4" else "0")  # w4: the 4-wave early-release form of it
    else:                  # the persistent tile walk
        _needs_experimental(backend)
        monkeypatch.setenv("S2ST_GROUP_ONESHOT", "0")
        monkeypatch.setenv("S2ST_GROUP_TILE", str(tile))
    d = backend.device
    g = torch.Generator().manual_seed(21)
    T1, T2 = (200, 136) if backend.kind == "emu" else (4584, 3120)
    shapes = [(256, 128, T1), (128, 256, T1), (128, 128, T2), (384, 128, T2)]
    if tile == 256:
        shapes = [(256, 128, T1), (512, 256, T1), (320, 128, T2), (384, 384, T2)]
    keep, probs, refs = [], [], []
    for (N_out, K_in, T) in shapes:
        dY, X = _bf(torch.randn(T, N_out, generator=g)), _bf(torch.randn(T, K_in, generator=g))
        dW0 = torch.randn(N_out, K_in, generator=g)
        dW = dW0.clone().to(d)
        dYd, Xd = dY.to(d), X.to(d)
        keep += [dYd, Xd, dW]
        probs.append(backend.bd.gemm_args_bf16(dYd, Xd, dW, N_out, K_in, T, a_kmajor=False, a_ld=N_out, b_kmajor=False,
                                               b_ld=K_in, accumulate=True))
        refs.append((dW, dW0.double() + dY.double().t() @ X.double()))
    backend.bd.gemm_group(probs)
    backend.sync()
    for dW, ref in refs:
        assert _relerr(dW, ref) < 2e-6


@pytest.mark.parametrize("akm,bkm", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("shape", ["wide", "narrow"])
def test_bf16_stream_k(backend, monkeypatch, akm, bkm, shape):
    """Stream-K form of the persistent kernel: the K-steps of all tiles are dealt out evenly, tiles that straddle two (or
    more) workgroups are completed through partial accumulators in a bound scratch buffer.  Against the exact product and
    against the unsplit launch (same products, different fp32 summation order); two launches in a row (the ticket
    counters re-arm themselves)."""
    import ctypes as C
    _needs_experimental(backend)
    monkeypatch.setenv("S2ST_GEMM_PERSIST", "1")
    monkeypatch.setenv("S2ST_STREAMK_MIN_STEPS", "2")
    monkeypatch.setenv("S2ST_GEMM_TILE", "128x128")
    emu = backend.kind == "emu"
    # wide: more tiles than workgroups, uneven rounds; narrow: fewer tiles than workgroups, long K (a tile spans several
    # workgroups)
    M, N, K = ((640, 256, 200) if emu else (4584, 2048, 512)) if shape == "wide" else ((384, 128, 1536) if emu else (4584, 512, 2048))
    g = torch.Generator().manual_seed(K + 2 * akm + bkm)
    A, B = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g))
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Am, a_ld = _pad_cols(A if akm else A.t().contiguous())
    Bm, b_ld = _pad_cols(B if bkm else B.t().contiguous())
    d = backend.device
    lib = backend.bd.lib()
    lib.s2st_gemm_streamk_scratch_floats.restype = C.c_int64
    lib.s2st_gemm_streamk_scratch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    scratch = torch.zeros(int(lib.s2st_gemm_streamk_scratch_floats()), device=d)
    R = A.double() @ B.double().t()
    ref = torch.relu(0.5 * R + bias.double()) + res.double()
    kw = dict(a_kmajor=akm, b_kmajor=bkm, a_ld=a_ld, b_ld=b_ld, alpha=0.5, bias=bias.to(d), act=1, resid=res.to(d))
    C0 = torch.zeros(M, N, device=d)
    backend.bd.gemm(Am.to(d), Bm.to(d), C0, M, N, K, **kw)  # unsplit
    backend.sync()
    assert lib.s2st_gemm_streamk_scratch(scratch.data_ptr(), scratch.numel(), C.c_void_p(backend.bd.stream_ptr())) == 0
    try:
        outs = []
        for _ in range(2):
            C1 = torch.full((M, N), 7.0, device=d)
            Ch = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
            backend.bd.gemm(Am.to(d), Bm.to(d), C1, M, N, K, c_bf16=Ch, **kw)
            backend.sync()
            outs.append(C1)
            assert _relerr(C1, ref) < 2e-6
            assert torch.equal(Ch.cpu(), C1.cpu().to(torch.bfloat16))
        neq = outs[0] != outs[1]  # fixed order of the partial sums: run-to-run identical
        assert not bool(neq.any()), "%d elements differ between two runs, max %.3e, first at %s" % (
            int(neq.sum()), float((outs[0] - outs[1]).abs().max()), neq.nonzero()[:4].tolist())
        assert _relerr(outs[0], C0.double().cpu()) < 3e-6  # (fp32 sums of K = 2048 products in two groupings)
        ctr = scratch.view(torch.int32)[:9]
        assert int(ctr.abs().sum()) == 0, "ticket / completion counters must be re-armed by the last workgroup"
        assert int(scratch.view(torch.int32)[16:16 + 512].max()) > 0, "no workgroup handed over a partial tile: stream-K did not run"
    finally:
        lib.s2st_gemm_streamk_scratch(None, 0, C.c_void_p(backend.bd.stream_ptr()))
