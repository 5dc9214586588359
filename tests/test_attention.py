"""Fused attention kernels (attention.hip, through the C ABI) against a float64 restatement of
fairseq MultiheadAttention's core (multihead_attention.py:224-367): masks, softmax, P*V and its
autograd.  Inputs are bf16-rounded; probabilities / score gradients pass through bf16 inside the
kernels, hence the 1-2 % tolerances."""
import pytest
import torch


def reference(q, k, v, H, klen, causal, dO=None):
    B, T, Cm = q.shape
    S = k.shape[1]
    dh = Cm // H
    q, k, v = (x.double().requires_grad_(True) for x in (q, k, v))
    qh = q.view(B, T, H, dh).permute(0, 2, 1, 3) * dh ** -0.5
    kh = k.view(B, S, H, dh).permute(0, 2, 1, 3)
    vh = v.view(B, S, H, dh).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2)
    mask = torch.zeros(B, 1, T, S, dtype=torch.bool)
    if klen is not None:
        mask |= (torch.arange(S)[None, :] >= klen[:, None])[:, None, None, :]
    if causal:
        mask |= (torch.arange(S)[None, :] > torch.arange(T)[:, None])[None, None]
    s = s.masked_fill(mask, float("-inf"))
    p = torch.softmax(s, -1)
    o = (p @ vh).permute(0, 2, 1, 3).reshape(B, T, Cm)
    lse = torch.logsumexp(s, -1)
    if dO is None:
        return o.detach(), lse.detach()
    o.backward(dO.double())
    return o.detach(), lse.detach(), q.grad, k.grad, v.grad


def rel(x, y):
    return float((x.double().cpu() - y).abs().max() / (y.abs().max() + 1e-12))


@pytest.mark.parametrize("dh,H", [(64, 2), (128, 1)])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("bf16_o", [False, True], ids=["dvec_kernel", "d_in_kernel"])
def test_flash_attention_fwd_bwd(backend, dh, H, causal, bf16_o):
    B, T, S = 2, 37, 37 if causal else 45
    Cm = H * dh
    g = torch.Generator().manual_seed(dh + 7 * causal)
    q = torch.randn(B, T, Cm, generator=g).to(torch.bfloat16)
    k = torch.randn(B, S, Cm, generator=g).to(torch.bfloat16)
    v = torch.randn(B, S, Cm, generator=g).to(torch.bfloat16)
    dO = torch.randn(B, T, Cm, generator=g)
    klen = torch.tensor([S, S - 9], dtype=torch.int32)
    d = backend.device
    o, lse, dq, dk, dv = backend.bd.flash_attention(q.to(d), k.to(d), v.to(d), H, klen=klen.to(d), causal=causal,
                                                    dO=dO.to(d), bf16_o=bf16_o)
    backend.sync()
    ro, rl, rq, rk, rv = reference(q.float(), k.float(), v.float(), H, klen.long(), causal,
                                   dO.to(torch.bfloat16).float())
    assert rel(o, ro) < 1e-2
    assert rel(lse, rl) < 1e-4
    assert rel(dq, rq) < 2e-2 and rel(dk, rk) < 2e-2 and rel(dv, rv) < 2e-2
    # keys beyond klen get no gradient
    assert float(dk[1, S - 9:].abs().max()) == 0.0 and float(dv[1, S - 9:].abs().max()) == 0.0
    # bf16 gradient copies + fused bias-gradient column sums
    _, _, dq2, dk2, dv2, (dqh, dkh, dvh, dbq, dbk, dbv) = backend.bd.flash_attention(
        q.to(d), k.to(d), v.to(d), H, klen=klen.to(d), causal=causal, dO=dO.to(d), bf16_grads=True, bf16_o=bf16_o)
    backend.sync()
    for full, half, db in ((dq2, dqh, dbq), (dk2, dkh, dbk), (dv2, dvh, dbv)):
        assert torch.equal(half.cpu(), full.cpu().to(torch.bfloat16))
        ref_db = full.cpu().double().sum(dim=(0, 1))
        assert float((db.cpu().double() - ref_db).abs().max()) < 1e-4 * float(ref_db.abs().max() + 1e-6)


def test_flash_attention_xcd_block_order(backend):
    """Ten (batch, head) pairs: one full group of 8 (whose blocks are re-dealt so that a pair's blocks share an XCD) and two
    pairs in launch order; forward, backward and the bias-gradient partial sums against the reference."""
    B, H, dh, T, S = 5, 2, 64, 70, 83
    Cm = H * dh
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, T, Cm, generator=g).to(torch.bfloat16)
    k = torch.randn(B, S, Cm, generator=g).to(torch.bfloat16)
    v = torch.randn(B, S, Cm, generator=g).to(torch.bfloat16)
    dO = torch.randn(B, T, Cm, generator=g)
    klen = torch.tensor([S, S - 9, 40, S, 64], dtype=torch.int32)
    d = backend.device
    o, lse, dq, dk, dv, (dqh, dkh, dvh, dbq, dbk, dbv) = backend.bd.flash_attention(
        q.to(d), k.to(d), v.to(d), H, klen=klen.to(d), dO=dO.to(d), bf16_grads=True, bf16_o=True)
    backend.sync()
    ro, rl, rq, rk, rv = reference(q.float(), k.float(), v.float(), H, klen.long(), False, dO.to(torch.bfloat16).float())
    assert rel(o, ro) < 1e-2 and rel(lse, rl) < 1e-4
    assert rel(dq, rq) < 2e-2 and rel(dk, rk) < 2e-2 and rel(dv, rv) < 2e-2
    for full, db in ((dq, dbq), (dk, dbk), (dv, dbv)):
        ref_db = full.cpu().double().sum(dim=(0, 1))
        assert float((db.cpu().double() - ref_db).abs().max()) < 1e-4 * float(ref_db.abs().max() + 1e-6)


def test_flash_attention_dropout_matches_unfused_mask(backend):
    """Same (seed, element) -> same keep decision as the unfused softmax kernel: o_fused == dropout(p) v."""
    B, H, T, S, dh = 1, 1, 20, 24, 64
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, T, dh, generator=g).to(torch.bfloat16)
    k = torch.randn(B, S, dh, generator=g).to(torch.bfloat16)
    v = torch.randn(B, S, dh, generator=g).to(torch.bfloat16)
    d = backend.device
    o, _ = backend.bd.flash_attention(q.to(d), k.to(d), v.to(d), H, drop_p=0.3, seed=99)
    sc = (q.float() @ k.float().transpose(1, 2)) * dh ** -0.5
    ld = 24
    s_buf = sc.view(1, 1, T, S).contiguous().to(d)
    p = torch.zeros(1, 1, T, ld, device=d)
    pd = torch.zeros(1, 1, T, ld, device=d)
    backend.bd.call("s2st_softmax_fwd_f32", s_buf, p, pd, None, 1, 1, T, S, ld, 0, 0.3, 99)
    backend.sync()
    ref = pd[0, 0, :, :S].cpu().double() @ v[0].double()
    assert rel(o[0], ref) < 1e-2


# the BENCH's head geometry (dh 128 x H 4) at the lengths its batches really have -- one partial
# tile (31), exact tile edges (32, 64, 128), one past them (33, 129), several key blocks (213 = the 16 x 850 batch's encoder
# length, 349 = the longest timed batch), the corpus maximum (750 encoder positions, s2st_transformer.py max-source-positions
# 3000 / 4) -- self-attention with and without the causal mask (causal tile skipping needs T > 128) and cross-attention
# rectangles (decoder steps x encoder length of the timed batches); key lengths exactly ON the 32 / 64 / 128 boundaries and
# klen = 1; 16 (batch, head) pairs so that the XCD re-deal of the blocks engages at dh 128.
# Reference semantics: multihead_attention.py:332-367, s2st_transformer.py:465-477 (future mask).
BENCH_GEOMS = [(31, 31), (32, 32), (33, 33), (64, 64), (127, 127), (128, 128), (129, 129), (213, 213), (349, 349),
               (750, 750), (72, 108), (140, 213), (219, 349)]


def _boundary_klens(S):
    edges = [e for e in (128, 64, 32) if e < S]
    ks = [S] + edges[:2]
    while len(ks) < 3:
        ks.append(max(S - 1, 1))
    return ks + [1]


@pytest.mark.parametrize("causal", [False, True], ids=["full", "causal"])
@pytest.mark.parametrize("T,S", BENCH_GEOMS, ids=[f"{t}x{s}" for t, s in BENCH_GEOMS])
def test_flash_attention_bench_head_geometry(backend, T, S, causal):
    if causal and T != S:
        pytest.skip("the causal mask belongs to self-attention (T == S)")
    if backend.kind == "emu" and (T, S) not in ((31, 31), (33, 33), (129, 129)):
        pytest.skip("the emulator runs three small geometries; the sweep runs on the GPU (the CPU suite's time budget)")
    H, dh = 4, 128
    klen = torch.tensor(_boundary_klens(S), dtype=torch.int32)
    B = klen.numel()
    if backend.kind == "emu":
        B, klen = 2, klen[[1, 3]] if S > 32 else klen[[0, 3]]
    Cm = H * dh
    g = torch.Generator().manual_seed(1000 * T + S + causal)
    q = torch.randn(B, T, Cm, generator=g).to(torch.bfloat16)
    k = torch.randn(B, S, Cm, generator=g).to(torch.bfloat16)
    v = torch.randn(B, S, Cm, generator=g).to(torch.bfloat16)
    dO = torch.randn(B, T, Cm, generator=g)
    d = backend.device
    o, lse, dq, dk, dv, (dqh, dkh, dvh, dbq, dbk, dbv) = backend.bd.flash_attention(
        q.to(d), k.to(d), v.to(d), H, klen=klen.to(d), causal=causal, dO=dO.to(d), bf16_grads=True, bf16_o=True)
    backend.sync()
    ro, rl, rq, rk, rv = reference(q.float(), k.float(), v.float(), H, klen.long(), causal, dO.to(torch.bfloat16).float())
    assert rel(o, ro) < 1e-2, rel(o, ro)
    assert rel(lse, rl) < 1e-4
    assert rel(dq, rq) < 2e-2 and rel(dk, rk) < 2e-2 and rel(dv, rv) < 2e-2, (rel(dq, rq), rel(dk, rk), rel(dv, rv))
    for b in range(B):  # keys beyond klen get no gradient, and nothing leaks into them
        kl = int(klen[b])
        if kl < S:
            assert float(dk[b, kl:].abs().max()) == 0.0 and float(dv[b, kl:].abs().max()) == 0.0, b
    for full, half, db in ((dq, dqh, dbq), (dk, dkh, dbk), (dv, dvh, dbv)):
        assert torch.equal(half.cpu(), full.cpu().to(torch.bfloat16))
        ref_db = full.cpu().double().sum(dim=(0, 1))
        assert float((db.cpu().double() - ref_db).abs().max()) < 1e-4 * float(ref_db.abs().max() + 1e-6)
