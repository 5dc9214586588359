"""Per-kernel parity: every C-ABI op against a plain PyTorch fp32 reference / the oracle.
Runs on the emulator build (CPU, `-m "not gpu"`) and on the product library (`-m gpu`)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import s2st_oracle as O


def dev(b, *ts):
    return [t.to(b.device) if t is not None else None for t in ts]


def close(a, b, rtol=1e-5, atol=1e-6, msg=""):
    np.testing.assert_allclose(a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy(),
                               rtol=rtol, atol=atol, err_msg=msg)


@pytest.mark.parametrize("rows,cols", [(37, 128), (5, 64), (130, 512), (9, 768)])
def test_layernorm(backend, rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g) * 2 + 0.5
    gam, bet, dy = torch.randn(cols, generator=g), torch.randn(cols, generator=g), torch.randn(rows, cols, generator=g)
    xd, gd, bd_, dyd = dev(backend, x, gam, bet, dy)
    y = torch.empty_like(xd)
    mean = torch.empty(rows, device=backend.device)
    rstd = torch.empty(rows, device=backend.device)
    backend.bd.call("s2st_layernorm_fwd_f32", xd, gd, bd_, y, mean, rstd, rows, cols, 1e-5)
    xr = x.clone().requires_grad_()
    gr, br = gam.clone().requires_grad_(), bet.clone().requires_grad_()
    ref = F.layer_norm(xr, (cols,), gr, br, 1e-5)
    ref.backward(dy)
    backend.sync()
    close(y, ref, 1e-5, 1e-5)
    dx = torch.ones_like(xd)
    dg = torch.zeros(cols, device=backend.device)
    db = torch.zeros(cols, device=backend.device)
    scratch = torch.zeros(backend.bd._bind("s2st_layernorm_bwd_scratch")(rows, cols), device=backend.device)
    backend.bd.call("s2st_layernorm_bwd_f32", dyd, xd, gd, mean, rstd, dx, 1, dg, db, scratch, rows, cols)
    backend.sync()
    close(dx, xr.grad + 1.0, 1e-4, 1e-5)
    close(dg, gr.grad, 1e-4, 1e-4)
    close(db, br.grad, 1e-4, 1e-4)


@pytest.mark.parametrize("causal", [0, 1])
def test_softmax_masks(backend, causal):
    B, H, T, S, ld = 3, 2, 21, 21 if causal else 70, 72
    g = torch.Generator().manual_seed(5)
    s = torch.randn(B, H, T, ld, generator=g) * 3
    klen = torch.tensor([S, 5, 13], dtype=torch.int32)
    dp = torch.randn(B, H, T, ld, generator=g)
    sd, kd, dpd = dev(backend, s, klen, dp)
    p = torch.zeros_like(sd)
    backend.bd.call("s2st_softmax_fwd_f32", sd, p, None, kd, B, H, T, S, ld, causal, 0.0, 0)
    sr = s[..., :S].clone().requires_grad_()
    m = torch.zeros(B, 1, T, S)
    for b in range(B):
        m[b, :, :, klen[b]:] = float("-inf")
    if causal:
        m = m + torch.triu(torch.full((T, S), float("-inf")), 1)
    ref = torch.softmax(sr + m, -1)
    ref.backward(dp[..., :S])
    backend.sync()
    close(p[..., :S], ref, 1e-5, 1e-6)
    ds = torch.zeros_like(sd)
    backend.bd.call("s2st_softmax_bwd_f32", p, dpd, ds, B, H, T, S, ld, 0.0, 0)
    backend.sync()
    close(ds[..., :S], sr.grad, 1e-4, 1e-6)


def test_softmax_dropout_consistency(backend):
    B, H, T, S, ld = 2, 2, 16, 40, 40
    s = torch.randn(B, H, T, ld)
    sd, = dev(backend, s)
    p, pd = torch.zeros_like(sd), torch.zeros_like(sd)
    backend.bd.call("s2st_softmax_fwd_f32", sd, p, pd, None, B, H, T, S, ld, 0, 0.3, 77)
    backend.sync()
    mask = (pd != 0).float() / 0.7
    close(pd, p * mask, 1e-6, 1e-7)
    assert abs((pd != 0).float().mean().item() - 0.7) < 0.05
    # backward regenerates the same mask
    dpd = torch.randn(B, H, T, ld, device=backend.device)
    ds = torch.zeros_like(sd)
    backend.bd.call("s2st_softmax_bwd_f32", p, dpd, ds, B, H, T, S, ld, 0.3, 77)
    backend.sync()
    dpp = dpd * mask
    ref = p * (dpp - (dpp * p).sum(-1, keepdim=True))
    close(ds, ref, 1e-4, 1e-6)


def test_colsum_headmean(backend):
    x = torch.randn(333, 70)
    xd, = dev(backend, x)
    out = torch.ones(70, device=backend.device)
    backend.bd.call("s2st_colsum_f32", xd, 70, 333, 70, out, 1)
    backend.sync()
    close(out, 1 + x.sum(0), 1e-5, 1e-4)
    p = torch.rand(2, 4, 7, 12)
    pd, = dev(backend, p)
    o = torch.zeros(2, 9, 7, device=backend.device)
    backend.bd.call("s2st_attn_headmean_f32", pd, o, 2, 4, 7, 9, 12)
    backend.sync()
    close(o, p[..., :9].mean(1).transpose(1, 2), 1e-6, 1e-7)


def test_glu_and_copy_rows_with_halo(backend):
    Bn, T, Cc = 3, 11, 8
    a = torch.randn(Bn * T, 2 * Cc)
    dy = torch.randn(Bn, T + 4, Cc)
    ad, dyd = dev(backend, a, dy)
    y = torch.zeros(Bn, T + 4, Cc, device=backend.device)
    sp = backend.bd.make_split(Cc, T, (T + 4) * Cc)
    backend.bd.call("s2st_glu_fwd_f32", ad, y.view(-1)[2 * Cc:], sp, Bn * T, Cc)
    ar = a.clone().requires_grad_()
    ref = F.glu(ar, dim=1)
    backend.sync()
    close(y[:, 2:T + 2].reshape(Bn * T, Cc), ref, 1e-5, 1e-6)
    assert float(y[:, :2].abs().sum()) == 0 and float(y[:, T + 2:].abs().sum()) == 0
    ref.backward(dy[:, 2:T + 2].reshape(Bn * T, Cc))
    da = torch.zeros_like(ad)
    backend.bd.call("s2st_glu_bwd_f32", ad, dyd.view(-1)[2 * Cc:], sp, da,
                    backend.bd.make_split(2 * Cc), Bn * T, Cc)
    backend.sync()
    close(da, ar.grad, 1e-5, 1e-6)
    # copy_rows: plain -> halo
    x = torch.randn(Bn * T, Cc)
    xd, = dev(backend, x)
    h = torch.zeros(Bn, T + 4, Cc, device=backend.device)
    backend.bd.call("s2st_copy_rows_f32", xd, backend.bd.make_split(Cc), h.view(-1)[2 * Cc:], sp, Bn * T, Cc)
    backend.sync()
    close(h[:, 2:T + 2].reshape(Bn * T, Cc), x, 0, 0)


def test_add_pe_embed_dropout(backend):
    rows, Cc = 40, 32
    x = torch.randn(rows, Cc)
    lens = torch.tensor([10, 7, 10, 3])
    pad = O.lengths_to_padding_mask(lens, 10)
    pos = O.make_positions(pad, 1).view(-1).to(torch.int32)
    table = O.sinusoidal_table(16, Cc, 1)
    alpha = torch.tensor([1.25])
    xd, posd, td, ald = dev(backend, x, pos, table, alpha)
    y = torch.zeros_like(xd)
    backend.bd.call("s2st_add_pe_f32", xd, y, posd, td, rows, Cc, 2.0, ald, 0.0, 0)
    ref = 2.0 * x + 1.25 * O.positional_embedding(pad, Cc).view(rows, Cc)
    backend.sync()
    close(y, ref, 1e-6, 1e-6)
    dy = torch.randn(rows, Cc)
    dyd, = dev(backend, dy)
    dal = torch.zeros(1, device=backend.device)
    backend.bd.call("s2st_pe_alpha_bwd_f32", dyd, posd, td, rows, Cc, 0.0, 0, dal)
    backend.sync()
    close(dal, (dy * O.positional_embedding(pad, Cc).view(rows, Cc)).sum().view(1), 1e-4, 1e-4)
    # embedding
    tok = torch.randint(0, 12, (rows,))
    tok[3] = 1
    emb = torch.randn(12, Cc)
    tokd, embd = dev(backend, tok, emb)
    e = torch.zeros(rows, Cc, device=backend.device)
    backend.bd.call("s2st_embed_fwd_f32", tokd, embd, e, rows, Cc, 3.0)
    backend.sync()
    close(e, 3.0 * emb[tok], 1e-6, 1e-6)
    de = torch.zeros(12, Cc, device=backend.device)
    backend.bd.call("s2st_embed_bwd_f32", tokd, dyd, de, rows, Cc, 3.0, 1)
    backend.sync()
    ref = torch.zeros(12, Cc).index_add_(0, tok, 3.0 * dy)
    ref[1] = 0
    close(de, ref, 1e-5, 1e-5)
    # ... and the ordered form (what the engine runs): more rows than one 4096-row window, ids without any row, twice
    big = 4096 + 777
    tokb = torch.randint(0, 9, (big,))
    tokb[tokb == 5] = 6  # id 5 has no row
    dyb = torch.randn(big, Cc)
    tokbd, dybd = dev(backend, tokb, dyb)
    refb = torch.zeros(12, Cc, dtype=torch.float64).index_add_(0, tokb, 3.0 * dyb.double())
    refb[1] = 0
    got = []
    for _ in range(2):
        de = torch.zeros(12, Cc, device=backend.device)
        backend.bd.call("s2st_embed_bwd_ordered_f32", tokbd, dybd, de, big, Cc, 12, 3.0, 1)
        backend.sync()
        got.append(de.clone())
    close(got[0], refb.float(), 1e-5, 1e-4)
    assert torch.equal(got[0], got[1]) and not bool(got[0][5].any())
    # dropout: same seed same mask, scale a, accumulate
    y1 = torch.zeros(rows * Cc, device=backend.device)
    y2 = torch.ones(rows * Cc, device=backend.device)
    backend.bd.call("s2st_dropout_f32", xd, y1, rows * Cc, 2.0, 0.5, 9, 0)
    backend.bd.call("s2st_dropout_f32", xd, y2, rows * Cc, 2.0, 0.5, 9, 1)
    backend.sync()
    close(y2, y1 + 1, 1e-6, 1e-6)
    kept = (y1 != 0).cpu()
    close(y1.cpu()[kept], (4.0 * x.view(-1))[kept], 1e-6, 1e-6)
    assert 0.4 < kept.float().mean() < 0.6


@pytest.mark.parametrize("tanh_", [1, 0])
def test_batchnorm_train(backend, tanh_):
    rows, Cc = 96, 40
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, Cc, generator=g) * 1.5 + 0.3
    gam, bet = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    dy, res = torch.randn(rows, Cc, generator=g), torch.randn(rows, Cc, generator=g)
    bn = torch.nn.BatchNorm1d(Cc)
    with torch.no_grad():
        bn.weight.copy_(gam)
        bn.bias.copy_(bet)
    bn.train()
    xr = x.clone().requires_grad_()
    u = bn(xr)
    ref = (torch.tanh(u) if tanh_ else u) + res
    ref.backward(dy)
    xd, gd, bd_, dyd, resd = dev(backend, x, gam, bet, dy, res)
    mean, var = torch.zeros(Cc, device=backend.device), torch.zeros(Cc, device=backend.device)
    rm, rv = torch.zeros(Cc, device=backend.device), torch.ones(Cc, device=backend.device)
    tmp = torch.full((130 * Cc,), float("nan"), device=backend.device)  # S2ST_BN_TMP_FLOATS(C)
    backend.bd.call("s2st_bn_stats_f32", xd, rows, Cc, mean, var, rm, rv, 0.1, tmp)
    y = torch.zeros_like(xd)
    sp = backend.bd.make_split(Cc)
    backend.bd.call("s2st_bn_apply_f32", xd, mean, var, gd, bd_, y, sp, resd, rows, Cc, 1e-5, tanh_, 0.0, 0)
    backend.sync()
    close(rm, bn.running_mean, 1e-5, 1e-6)
    close(rv, bn.running_var, 1e-5, 1e-6)
    close(y, ref, 1e-5, 1e-5)
    dx = torch.zeros_like(xd)
    dg, db = torch.zeros(Cc, device=backend.device), torch.zeros(Cc, device=backend.device)
    backend.bd.call("s2st_bn_bwd_f32", dyd, sp, xd, mean, var, gd, bd_, dx, sp, dg, db, tmp, rows, Cc,
                    1e-5, tanh_, 0.0, 0)
    backend.sync()
    close(dx, xr.grad, 1e-4, 1e-5)
    close(dg, bn.weight.grad, 1e-4, 1e-4)
    close(db, bn.bias.grad, 1e-4, 1e-4)


def test_conv_weight_layouts(backend):
    Oc, Ic, Kw = 6, 4, 5
    w = torch.randn(Oc, Ic, Kw)
    wd_, = dev(backend, w)
    wf = torch.zeros(Oc, Kw, Ic, device=backend.device)
    wdg = torch.zeros(Ic, Kw, Oc, device=backend.device)
    backend.bd.call("s2st_conv_w_permute_f32", wd_, wf, wdg, Oc, Ic, Kw)
    backend.sync()
    close(wf, w.permute(0, 2, 1), 0, 0)
    close(wdg, w.flip(2).permute(1, 2, 0), 0, 0)
    dw = torch.ones(Oc, Ic, Kw, device=backend.device)
    backend.bd.call("s2st_conv_w_unpermute_acc_f32", wf, dw, Oc, Ic, Kw)
    backend.sync()
    close(dw, 1 + w, 0, 0)


def test_mel_loss(backend):
    B, D, Fd = 4, 9, 20
    g = torch.Generator().manual_seed(0)
    feat, post, tgt = (torch.randn(B, D, Fd, generator=g) for _ in range(3))
    eos = torch.randn(B, D, generator=g) * 2
    lens = torch.tensor([9, 4, 1, 6], dtype=torch.int32)
    fr, pr, er = feat.clone().requires_grad_(), post.clone().requires_grad_(), eos.clone().requires_grad_()
    mask = ~O.lengths_to_padding_mask(lens.long(), D)
    et = (torch.arange(D).view(1, D) == (lens.long().view(B, 1) - 1)).float()
    l1 = F.l1_loss(fr[mask], tgt[mask]) + F.l1_loss(pr[mask], tgt[mask])
    mse = F.mse_loss(fr[mask], tgt[mask]) + F.mse_loss(pr[mask], tgt[mask])
    bce = F.binary_cross_entropy_with_logits(er[mask], et[mask], pos_weight=torch.tensor(5.0))
    (0.7 * l1 + 1.3 * mse + 0.9 * bce).backward()
    fd, pd, ed, td, ld_ = dev(backend, feat, post, eos, tgt, lens)
    stats = torch.zeros(3, device=backend.device)
    nr = int(lens.sum())
    nf = nr * Fd
    df, dp, de = torch.zeros_like(fd), torch.zeros_like(pd), torch.zeros_like(ed)
    backend.bd.call("s2st_mel_loss_f32", fd, pd, ed, td, ld_, B, D, Fd, 5.0, stats, 0.7 / nf, 1.3 / nf,
                    0.9 / nr, df, dp, de)
    backend.sync()
    st = stats.cpu()
    close(st[0] / nf, l1, 1e-5, 1e-6)
    close(st[1] / nf, mse, 1e-5, 1e-6)
    close(st[2] / nr, bce, 1e-5, 1e-6)
    close(df, fr.grad, 1e-4, 1e-7)
    close(dp, pr.grad, 1e-4, 1e-7)
    close(de, er.grad, 1e-4, 1e-7)


@pytest.mark.parametrize("V", [44, 74, 7])
def test_label_smoothed_ce(backend, V):
    rows = 50
    g = torch.Generator().manual_seed(V)
    logits = torch.randn(rows, V, generator=g) * 2
    tgt = torch.randint(0, V, (rows,), generator=g)
    tgt[::7] = 1
    lr = logits.clone().requires_grad_()
    lp = F.log_softmax(lr, -1)
    loss, nll = O.label_smoothed_nll_loss(lp, tgt, 0.1)
    (loss * 0.37).backward()
    m = tgt.ne(1)
    ncorr = int((lp.argmax(1)[m] == tgt[m]).sum())
    ld_, td = dev(backend, logits, tgt)
    stats = torch.zeros(4, device=backend.device)
    dl = torch.zeros_like(ld_)
    backend.bd.call("s2st_ls_ce_f32", ld_, td, rows, V, 1, 0.1, stats, dl, 0.37)
    backend.sync()
    st = stats.cpu()
    eps_i = 0.1 / (V - 1)
    close((1 - 0.1 - eps_i) * st[0] + eps_i * st[1], loss, 1e-5, 1e-5)
    close(st[0], nll, 1e-5, 1e-5)
    assert int(st[2]) == ncorr and int(st[3]) == int(m.sum())
    close(dl, lr.grad, 1e-4, 1e-6)


def test_label_smoothing_reference_kat(backend, golden_dir):
    """Probability table from the reference's tests/test_label_smoothing.py."""
    import os
    z = np.load(os.path.join(golden_dir, "label_smoothing_kat.npz"))
    logits = torch.from_numpy(z["probs"]).log()
    tgt = torch.from_numpy(z["target"])
    ld_, td = dev(backend, logits, tgt)
    for eps in (0.0, 0.1, 0.3):
        stats = torch.zeros(4, device=backend.device)
        backend.bd.call("s2st_ls_ce_f32", ld_, td, 3, 7, 1, eps, stats, None, 1.0)
        backend.sync()
        st = stats.cpu().double()
        eps_i = eps / 6
        np.testing.assert_allclose([float((1 - eps - eps_i) * st[0] + eps_i * st[1]), float(st[0])],
                                   z[f"eps{eps}"], rtol=1e-5)


def test_batchnorm_statistics_with_a_large_mean(backend):
    """One-pass statistics (shifted sums): columns whose mean is 100 standard deviations away from zero -- where
    E[x^2] - E[x]^2 would lose every digit -- still give the variance of the two-pass formula."""
    rows, Cc = 1537, 24
    g = torch.Generator().manual_seed(5)
    x = torch.randn(rows, Cc, generator=g) * torch.linspace(0.5, 2.0, Cc) + torch.linspace(-300.0, 300.0, Cc)
    xd, = dev(backend, x)
    mean, var = torch.zeros(Cc, device=backend.device), torch.zeros(Cc, device=backend.device)
    rm, rv = torch.zeros(Cc, device=backend.device), torch.ones(Cc, device=backend.device)
    tmp = torch.full((130 * Cc,), float("nan"), device=backend.device)
    backend.bd.call("s2st_bn_stats_f32", xd, rows, Cc, mean, var, rm, rv, 0.1, tmp)
    backend.sync()
    xd64 = x.double()
    close(mean, xd64.mean(0).float(), 1e-6, 1e-5)
    close(var, xd64.var(0, unbiased=False).float(), 2e-5, 1e-7)


def test_ctc(backend):
    torch.manual_seed(0)
    B, E, V, Lmax = 5, 40, 11, 30
    logits = torch.randn(B, E, V)
    tl = torch.tensor([7, 1, 12, 3, 30], dtype=torch.int32)
    il = torch.tensor([40, 9, 33, 3, 31], dtype=torch.int32)
    tg = torch.randint(1, V, (B, Lmax))
    tg[4] = 3  # 30 repeats need 59 frames > 31: infeasible -> zero_infinity
    lr = logits.clone().requires_grad_()
    lp = F.log_softmax(lr, -1).transpose(0, 1)
    flat = torch.cat([tg[b, :tl[b]] for b in range(B)])
    ref = F.ctc_loss(lp, flat, il.long(), tl.long(), reduction="mean", zero_infinity=True)
    (ref * 0.3).backward()
    ld_, tgd, ild, tld = dev(backend, logits, tg, il, tl)
    lpo = torch.zeros(B, E, V, device=backend.device)
    per = torch.zeros(B, device=backend.device)
    dl = torch.full((B, E, V), 7.0, device=backend.device)
    nws = backend.bd._bind("s2st_ctc_workspace")(B, E, Lmax)
    ws = torch.zeros(nws, device=backend.device)
    backend.bd.call("s2st_ctc_f32", ld_, tgd, Lmax, ild, tld, B, E, V, lpo, per, dl, 0.3 / B, ws)
    backend.sync()
    close(lpo.transpose(0, 1), lp, 1e-5, 1e-5)
    close(per.mean(), ref, 1e-5, 1e-6)
    assert float(per[4]) == 0.0
    close(dl, lr.grad, 2e-4, 2e-6)
    # integer output: greedy path is bit-exact
    assert torch.equal(lpo.argmax(-1).cpu(), lp.transpose(0, 1).argmax(-1))


@pytest.mark.parametrize("E,Lmax", [(170, 72), (300, 140), (540, 260)], ids=["S145", "S281", "S521"])
def test_ctc_long_labels_and_repeatability(backend, E, Lmax):
    """More than 128 extended states (a thread of the alpha / beta halves then owns 2, 4 or 16 states: the three
    instantiations), a vocabulary beyond one wave, an empty target, and the same launch twice: the gradient comes out of
    ordered per-label sums (no atomics), so it repeats bit for bit."""
    torch.manual_seed(1)
    B, V = 4, 70
    logits = torch.randn(B, E, V)
    tl = torch.tensor([Lmax, Lmax - 7, 2, 0], dtype=torch.int32)
    il = torch.tensor([E, E - 10, 5, 9], dtype=torch.int32)
    tg = torch.randint(1, V, (B, Lmax))
    lr = logits.clone().requires_grad_()
    lp = F.log_softmax(lr, -1).transpose(0, 1)
    flat = torch.cat([tg[b, :tl[b]] for b in range(B)])
    ref = F.ctc_loss(lp, flat, il.long(), tl.long(), reduction="mean", zero_infinity=True)
    ref.backward()
    ld_, tgd, ild, tld = dev(backend, logits, tg, il, tl)
    nws = backend.bd._bind("s2st_ctc_workspace")(B, E, Lmax)
    outs = []
    for _ in range(2):
        lpo = torch.zeros(B, E, V, device=backend.device)
        per = torch.zeros(B, device=backend.device)
        dl = torch.full((B, E, V), 7.0, device=backend.device)
        ws = torch.zeros(nws, device=backend.device)
        backend.bd.call("s2st_ctc_f32", ld_, tgd, Lmax, ild, tld, B, E, V, lpo, per, dl, 1.0 / B, ws)
        backend.sync()
        outs.append((per.clone(), dl.clone()))
    close(outs[0][0].mean(), ref, 2e-5, 1e-6)
    close(outs[0][1], lr.grad, 5e-4, 5e-6)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("parts", [False, True], ids=["scalar_sumsq", "partial_sums"])
def test_sumsq_adam(backend, parts):
    """parts: the gradient norm as per-block partial sums folded in index order inside the Adam kernel (no atomics:
    what the trainer uses) instead of one atomically accumulated float."""
    n = 10007 if not parts else 64 * 1024 + 5  # (64 blocks' worth of partials; torch's own fp32 norm drifts by 6e-5 at 3 M)
    nparts = int(backend.bd._bind("s2st_sumsq_parts_count")(n)) if parts else 0
    assert not parts or nparts > 1

    def sumsq(gd, ss):
        if parts:
            backend.bd.call("s2st_sumsq_parts_f32", gd, n, ss)
        else:
            ss.zero_()
            backend.bd.call("s2st_sumsq_f32", gd, n, ss)

    g_ = torch.Generator().manual_seed(1)
    p0, g0 = torch.randn(n, generator=g_), torch.randn(n, generator=g_) * 3
    pr = torch.nn.Parameter(p0.clone())
    opt = O.FairseqAdam([pr], weight_decay=0.01)
    pd, gd = dev(backend, p0.clone(), g0.clone())
    m, v = torch.zeros(n, device=backend.device), torch.zeros(n, device=backend.device)
    half = torch.full((1,), 0.5, device=backend.device)  # device-side part of the multiplier
    for step in range(1, 4):
        gcur = g0 * step
        pr.grad = gcur.clone() * 0.01  # multiply_grads(1/sample_size)
        gn_ref = O.clip_grad_norm_([pr], 0.5)
        opt.step(1e-2)
        gd.copy_(gcur)
        ss = torch.full((max(nparts, 1),), float("nan"), device=backend.device)  # (partials need no zeroing)
        gno = torch.zeros(1, device=backend.device)
        sumsq(gd, ss)
        ph = torch.zeros(n, dtype=torch.bfloat16, device=backend.device)
        skipped = torch.zeros(1, dtype=torch.int32, device=backend.device)
        backend.bd.call("s2st_adam_f32", pd, gd, m, v, n, ss, 0.02, half, 0.5, 1e-2, 0.9, 0.999, 1e-8, 0.01,
                        step, gno, ph, skipped, nparts, step % 2)
        backend.sync()
        assert int(skipped) == 0
        if step % 2:  # zero_grad = 1: the arena is handed back cleared
            assert not bool(gd.any())
        else:  # the scaled and clipped gradient
            close(gd, pr.grad, 1e-5, 1e-7)
        close(gno, gn_ref.view(1), 1e-5, 1e-6)
        close(pd, pr.detach(), 1e-5, 1e-6)
        assert torch.equal(ph, pd.to(torch.bfloat16))  # the fused bf16 copy == a cast of the new parameters
    # non-finite gradient norm: nothing is touched and the device counter says so (trainer.py:860-867)
    before = (pd.clone(), m.clone(), v.clone())
    gd[3] = float("inf")
    sumsq(gd, ss)
    backend.bd.call("s2st_adam_f32", pd, gd, m, v, n, ss, 0.02, half, 0.5, 1e-2, 0.9, 0.999, 1e-8, 0.01, 4, gno, ph,
                    skipped, nparts, 0)
    backend.sync()
    assert int(skipped) == 1 and not bool(torch.isfinite(gno).all())
    assert torch.equal(pd, before[0]) and torch.equal(m, before[1]) and torch.equal(v, before[2])
    assert bool(torch.isinf(gd[3]))
    backend.bd.call("s2st_adam_f32", pd, gd, m, v, n, ss, 0.02, half, 0.5, 1e-2, 0.9, 0.999, 1e-8, 0.01, 4, gno, ph,
                    skipped, nparts, 1)  # skipped again, but the gradients are cleared as asked
    backend.sync()
    assert int(skipped) == 2 and not bool(gd.any()) and torch.equal(pd, before[0])


def test_dropout_mask_independence(backend):
    """the 2-multiply 32-bit dropout hash (csrc/s2st_common.h: mix32) was only checked for its keep
    RATE.  Here the masks themselves: (1) neighbouring elements of one site, at lags 1, 2, 64 and the row width (what a
    correlated mask would hit first: the lanes of a wave and the rows of a tile), (2) the SAME elements at two sites of one
    step -- seeds as the engine derives them, seed * 0x100000001B3 + site * 0x9E3779B97F4A7C15 -- and at the same site of
    two consecutive steps, (3) the two seed words separately (a seed differing only in its high word must give an
    unrelated mask).  Statistic: the phi coefficient of the two keep indicators, |phi| < 4.5 / sqrt(n) (a 4.5-sigma bound
    under independence: ~7e-6 false-alarm rate per comparison), and each mask's keep rate within 4.5 sigma of 1 - p."""
    n = 1 << 20 if backend.kind == "hip" else 1 << 17
    p = 0.3
    x = torch.ones(n, device=backend.device)

    def mask(seed):
        y = torch.empty_like(x)
        backend.bd.call("s2st_dropout_f32", x, y, n, 1.0, p, seed & ((1 << 64) - 1), 0)
        backend.sync()
        return (y.cpu() != 0).double()

    def phi(a, b):
        a, b = a - a.mean(), b - b.mean()
        return float((a * b).mean() / (a.std(unbiased=False) * b.std(unbiased=False)))

    def site_seed(step_seed, site):
        return (step_seed * 0x100000001B3 + site * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)

    bound = 4.5 / n ** 0.5
    rate_bound = 4.5 * (p * (1 - p) / n) ** 0.5
    m0 = mask(site_seed(1000003, 1))
    assert abs(float(m0.mean()) - (1 - p)) < rate_bound
    for lag in (1, 2, 3, 64, 512, 2048):
        assert abs(phi(m0[:-lag], m0[lag:])) < 4.5 / (n - lag) ** 0.5, lag
    others = {"next site": site_seed(1000003, 2), "site 40": site_seed(1000003, 40), "next step": site_seed(2000006, 1),
              "low word + 1": site_seed(1000003, 1) + 1, "high word + 1": site_seed(1000003, 1) + (1 << 32),
              "seed 0": 0, "seed 1": 1}
    masks = {k: mask(v) for k, v in others.items()}
    for k, m in masks.items():
        assert abs(float(m.mean()) - (1 - p)) < rate_bound, k
        assert abs(phi(m0, m)) < bound, (k, phi(m0, m))
        assert abs(phi(m0[1:], m[:-1])) < bound, (k, "shifted")  # not an index-shifted copy either
    assert abs(phi(masks["seed 0"], masks["seed 1"])) < bound
    # a 2 x 2 x 2 check over three sites: every one of the 8 joint outcomes at its product probability
    a, b, c = m0, masks["next site"], masks["site 40"]
    for va in (0.0, 1.0):
        for vb in (0.0, 1.0):
            for vc in (0.0, 1.0):
                q = (p if va == 0 else 1 - p) * (p if vb == 0 else 1 - p) * (p if vc == 0 else 1 - p)
                got = float(((a == va) & (b == vb) & (c == vc)).double().mean())
                assert abs(got - q) < 4.5 * (q * (1 - q) / n) ** 0.5, (va, vb, vc, got, q)


def test_exchange_proxy_moves_bytes_and_changes_nothing(backend):
    """s2st_exchange_proxy_f32 (bench.py --exchange-proxy: the one-GPU stand-in for the gradient all-reduce's kernels): it
    copies the requested share of the bucket into the scratch range -- cyclically when the share exceeds the bucket -- and
    leaves the bucket (the gradients) untouched; workgroup count and pace only change how long it takes."""
    d = backend.device
    n = 40000
    g = torch.Generator().manual_seed(3)
    bucket = torch.randn(n, generator=g).to(d)
    ref = bucket.clone()
    for move_bytes, wgs, gbps in ((n * 4, 4, 1000.0), (int(1.75 * n * 4), 16, 50.0), (4096 * 16, 2, 300.0)):
        scratch = torch.zeros(n, device=d)
        backend.bd.call("s2st_exchange_proxy_f32", bucket, scratch, n, move_bytes, wgs, gbps)
        backend.sync()
        assert torch.equal(bucket, ref)
        moved = min(move_bytes // 16 * 4, n)  # floats of the bucket the share reaches (cyclic beyond the bucket's end)
        assert torch.equal(scratch[:moved], ref[:moved])
        assert float(scratch[moved:].abs().max()) == 0.0 if moved < n else True
