"""Round-5 kernels on the MI355X, each against the launches it replaces and against float64 torch on the same inputs:
the one-launch register-resident EDC term (csrc/edcone.hip), the light gamma sweep behind it, the forward pair transform
with per-signal factors / the adjoint with several slot-ordered inputs (csrc/fft.hip), the records pass on unscaled
transfer functions and the fused tail + head of the band bank's step (csrc/blocktf.hip)."""
import numpy as np
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from diffgfdn_amd import hip_ops
    return hip_ops


def _to_pairs(a):
    if a.shape[0] % 2:
        a = torch.cat([a, torch.zeros_like(a[:1])])
    return torch.stack((a[0::2], a[1::2]), dim=-1).contiguous()


def _from_pairs(a2, m):
    out = torch.empty((2 * a2.shape[0], a2.shape[1]), dtype=a2.dtype, device=a2.device)
    out[0::2], out[1::2] = a2[..., 0], a2[..., 1]
    return out[:m]


def _edc_case(gen, nb, B, G, n, R, start, lens):
    items, S_ = nb * B, nb * G
    decay = torch.exp(-torch.arange(n) / (n / 7.0))
    xd = (torch.randn(nb * R, n, generator=gen) * decay).to(DEV)
    tau = (torch.randn(S_, n, generator=gen) * decay).to(DEV)
    rgain = torch.randn(items, G, generator=gen).to(DEV)
    rows = torch.tensor([q * R + int(i) for q in range(nb) for i in torch.randperm(R, generator=gen)[:B]], device=DEV)
    tgt = (torch.randn(nb * R, n, generator=gen) * decay).to(DEV)
    return items, S_, xd, tau, rgain, rows, tgt


def _edc_reference64(xd, rows, tau, rgain, nb, B, G, start, lens, T_db, mw, inv, gscale):
    """losses.py:187-238 in float64 with autograd: per item (loss, dL/dx on the window, dL/drgain)"""
    items = rgain.shape[0]
    band = torch.arange(items, device=xd.device) // B
    rg = rgain.double().clone().requires_grad_(True)
    x = xd.double()[rows] + (rg[:, :, None] * tau.double().view(nb, G, -1)[band]).sum(1)
    x = x.detach().requires_grad_(True)
    li, total = [], 0.0
    for b in range(items):
        L = lens[b // B]
        w = x[b, start:start + L]
        edc = torch.flip(torch.cumsum(torch.flip(w * w, [0]), 0), [0])
        db = torch.clamp(10.0 * torch.log10(edc.abs() + 1.1920928955078125e-07), min=-200.0)
        m = (mw[b // B] if mw.dim() == 2 else mw)[:L].double()
        l = (m * (T_db[rows[b], :L].double() - db).abs()).sum() * inv
        li.append(l.detach())
        total = total + l * gscale
    total.backward()
    gx = x.grad
    dots = (gx[:, None, :] * tau.double().view(nb, G, -1)[band]).sum(-1)
    return torch.stack(li), gx, dots


@pytest.mark.parametrize("G,B,banded,masked", [(4, 4, False, True), (4, 6, True, True), (3, 4, False, False),
                                               (2, 2, True, True)])
def test_edc_term_in_one_launch(ops, G, B, banded, masked):
    """gfdn_edc_lin_one against (i) the three-launch form on samples formed on the fly (gfdn_edc_loss_pairs_lin) with the
    dot products of gfdn_lin_gain_dots and (ii) a float64 autograd evaluation of losses.py:187-238: loss per item, dL/dx on
    the window, the EDC part of dL/drgain -- windows that are no multiple of 4 (a partial last group), per-band windows,
    group counts that do and do not fill whole signal pairs."""
    gen = torch.Generator(device="cpu").manual_seed(100 * G + B)
    nb, n, R, start = 3, 20011, B + 2, 160
    lens = [9001, 15003, 19000] if banded else [18999] * nb
    Lmax = max(lens)
    items, S_, xd, tau, rgain, rows, tgt = _edc_case(gen, nb, B, G, n, R, start, lens)
    tau2 = _to_pairs(tau)
    T_db = torch.zeros(nb * R, Lmax, device=DEV)
    for q, L in enumerate(lens):
        T_db[q * R:(q + 1) * R, :L] = ops.edc_target(tgt[q * R:(q + 1) * R].contiguous(), start, L)
    item_len = torch.tensor(lens, dtype=torch.int32, device=DEV).repeat_interleave(B).contiguous() if banded else None
    mw = ((torch.rand(nb, Lmax, generator=gen) > 0.5).float().to(DEV) / 100.0) if masked else None
    if mw is not None and not banded:
        mw = mw[0].contiguous()
    inv, gs = 0.37, 10.0
    parts = torch.full((items * G, 3), 7.0, device=DEV)
    li, gx = ops.edc_lin_one(xd, rows, tau2, rgain, nb, n, start, Lmax, T_db, mw, inv, gs, True, trows=rows,
                             item_len=item_len, dots=parts, col=1)
    assert float((parts[:, 0] - 7.0).abs().max()) == 0.0 and float((parts[:, 2] - 7.0).abs().max()) == 0.0
    # (i) the three-launch form (needs an even number of receivers per band: both B of this test are)
    li3, g2 = ops.edc_loss_pairs_lin(xd, rows, tau2, rgain, nb, n, start, Lmax, T_db, mw, inv, gs, True, trows=rows,
                                     item_len=item_len)
    g3 = _from_pairs(g2, items)
    assert rel_err(li.cpu(), li3.cpu()) < 5e-6
    for q, L in enumerate(lens):
        sl = slice(q * B, (q + 1) * B)
        assert rel_err(gx[sl, :L].cpu(), g3[sl, start:start + L].cpu()) < 2e-5, q
    # (ii) float64
    mw64 = mw if mw is not None else torch.ones(Lmax, device=DEV)
    li64, gx64, dots64 = _edc_reference64(xd, rows, tau, rgain, nb, B, G, start, lens, T_db, mw64, inv, gs)
    assert rel_err(li.cpu(), li64.float().cpu()) < 5e-6
    for q, L in enumerate(lens):
        sl = slice(q * B, (q + 1) * B)
        assert rel_err(gx[sl, :L].cpu(), gx64[sl, start:start + L].float().cpu()) < 2e-5, q
    assert rel_err(parts[:, 1].view(items, G).cpu(), dots64.float().cpu()) < 2e-5
    # value-only call
    li_v, none = ops.edc_lin_one(xd, rows, tau2, rgain, nb, n, start, Lmax, T_db, mw, inv, gs, False, trows=rows,
                                 item_len=item_len)
    assert none is None and torch.equal(li_v, li)


def test_edc_term_in_one_launch_full_window(ops):
    """the north-star window (47 360 samples from sample 640 of 65 537) and the longest window the launch takes"""
    gen = torch.Generator(device="cpu").manual_seed(5)
    nb, B, G, n, R, start = 2, 4, 4, 65537, 5, 640
    for L in (47360, 49152):
        items, S_, xd, tau, rgain, rows, tgt = _edc_case(gen, nb, B, G, n, R, start, [L] * nb)
        tau2 = _to_pairs(tau)
        T_db = ops.edc_target(tgt, start, L)
        mw = ((torch.rand(L, generator=gen) > 0.5).float() / float(L)).to(DEV)
        parts = torch.zeros((items * G, 1), device=DEV)
        li, gx = ops.edc_lin_one(xd, rows, tau2, rgain, nb, n, start, L, T_db, mw, 1.0 / B, 10.0, True, trows=rows,
                                 dots=parts, col=0)
        li3, g2 = ops.edc_loss_pairs_lin(xd, rows, tau2, rgain, nb, n, start, L, T_db, mw, 1.0 / B, 10.0, True, trows=rows)
        g3 = _from_pairs(g2, items)[:, start:start + L]
        assert rel_err(li.cpu(), li3.cpu()) < 5e-6
        assert rel_err(gx.cpu(), g3.cpu()) < 2e-5
        dots3 = ops.tf_rows_sum(ops.lin_gain_dots(g2, tau2, nb, items, G, n, True, True)).view(items, G)
        assert rel_err(parts.view(items, G).cpu(), dots3.cpu()) < 2e-5
    assert not ops.edc_lin_one_supported(49153, 4) and ops.edc_lin_one_supported(49152, 4)


@pytest.mark.parametrize("banded,G", [(False, 4), (True, 4), (False, 3)])
def test_gamma_sweep_on_window_rows(ops, banded, G):
    """gfdn_lin_gamma_win (window-only plain rows in, slot order out, two bases) against gfdn_lin_gamma on zero-padded
    pair-interleaved signals; columns behind a band's window are poisoned: they must never be read."""
    gen = torch.Generator(device="cpu").manual_seed(29 + G)
    n, nb, B, w0 = 65537, 3, 6, 640
    items, S_ = nb * B, nb * G
    lens = [18561, 31360, 47360] if banded else [47360] * nb
    Lmax = max(lens)
    rgain = torch.randn(items, G, generator=gen).to(DEV)
    gx = torch.randn(items, Lmax, generator=gen).to(DEV)
    base = torch.randn((S_ + 1) // 2, n, 2, generator=gen).to(DEV)
    full = torch.zeros(items, n, device=DEV)
    gp = gx.clone()
    for q, L in enumerate(lens):
        full[q * B:(q + 1) * B, w0:w0 + L] = gx[q * B:(q + 1) * B, :L]
        gp[q * B:(q + 1) * B, L:] = float("nan")
    sot = ops.irfft_slot_of_time(n, torch.device(DEV))
    band_len = torch.tensor(lens, dtype=torch.int32, device=DEV) if banded else None
    for slots in (sot, None):
        ref = ops.lin_gamma(_to_pairs(full), rgain, nb, n, True, True, slot_of_time=slots, base=base)
        gam = ops.lin_gamma_win(gp, rgain, nb, n, w0, Lmax, base=base, slot_of_time=slots, band_win_len=band_len)
        if S_ % 2 == 0:
            assert rel_err(gam.cpu(), ref.cpu()) < 2e-6
        else:
            assert rel_err(gam[..., :].cpu()[:-1], ref.cpu()[:-1]) < 2e-6
            assert rel_err(gam[-1, :, 0].cpu(), ref[-1, :, 0].cpu()) < 2e-6
        half = (0.5 * base).contiguous()
        gam2 = ops.lin_gamma_win(gp, rgain, nb, n, w0, Lmax, base=half, base_b=half, slot_of_time=slots,
                                 band_win_len=band_len)
        assert rel_err(gam2.cpu(), gam.cpu()) < 1e-6
    # float64
    ref64 = (rgain.double().view(nb, B, G, 1) * full.double().view(nb, B, 1, n)).sum(1).reshape(S_, n)
    gam_t = ops.lin_gamma_win(gp, rgain, nb, n, w0, Lmax, band_win_len=band_len)
    assert rel_err(_from_pairs(gam_t, S_).cpu(), ref64.float().cpu()) < 5e-6


def test_pair_transform_with_signal_factors_and_summed_adjoint_inputs(ops):
    """gfdn_irfft_odd_pairs_fwd_scaled == factor x gfdn_irfft_odd_pairs_fwd (whole transform and split in front of the last
    pass); gfdn_irfft_odd_pairs_bwd_tslots3 of three slot-ordered parts == gfdn_irfft_odd_pairs_bwd_tslots of their sum."""
    gen = torch.Generator(device="cpu").manual_seed(17)
    n, batch = 65537, 6
    half = (n + 1) // 2
    X = torch.randn(batch, half, 2, generator=gen).to(DEV)
    X = torch.view_as_complex(X).contiguous()
    X[:, 0] = X[:, 0].real
    sc = (0.5 + torch.rand(batch, generator=gen)).to(DEV)
    x_ref = ops.irfft_odd_fwd(X, n, slots=True, pairs=True)
    want = _from_pairs(x_ref, batch) * sc[:, None]
    x_s = ops.irfft_odd_fwd(X, n, slots=True, pairs=True, oscale=sc)
    assert rel_err(_from_pairs(x_s, batch).cpu(), want.cpu()) < 1e-6
    hit = []
    x_t = ops.irfft_odd_fwd(X, n, slots=True, pairs=True, oscale=sc, before_last=lambda: hit.append(1))
    assert hit == [1] and torch.equal(x_t, x_s)
    parts = [torch.randn(batch // 2, n, 2, generator=gen).to(DEV) for _ in range(3)]
    one = ops.irfft_odd_pairs_bwd((parts[0] + parts[1]) + parts[2], n, batch, tslots=True)
    three = ops.irfft_odd_pairs_bwd(parts[0], n, batch, tslots=True, g2b=parts[1], g2c=parts[2])
    assert rel_err(three.cpu(), one.cpu()) < 1e-6
    two = ops.irfft_odd_pairs_bwd(parts[0], n, batch, tslots=True, g2b=parts[1])
    assert rel_err(two.cpu(), ops.irfft_odd_pairs_bwd(parts[0] + parts[1], n, batch, tslots=True).cpu()) < 1e-6


def _bank_blocks(gen, nblk, n):
    M = (torch.randn(nblk, n, n, generator=gen) / np.sqrt(n)).to(DEV)
    b = (torch.randn(nblk * n, generator=gen) / n).to(DEV)
    c = (torch.randn(nblk * n, generator=gen) / n).to(DEV)
    ig = (1.0 + 0.01 * torch.rand(nblk * n, generator=gen)).to(DEV)
    return M, b, c, ig


@pytest.mark.parametrize("n", [4, 3])
def test_records_pass_on_unscaled_transfer_functions(ops, n):
    """gfdn_tf_compose_bwd(tscale) on the UNSCALED saved functions == the pass on saved scaled ones (the normalisation scale
    joining behind the transform, bankstep.FusedBankStep.scale_late)."""
    gen = torch.Generator(device="cpu").manual_seed(7 + n)
    nb, G, K = 2, 4, 4097
    nblk = nb * G
    M, b, c, ig = _bank_blocks(gen, nblk, n)
    z = torch.tensor(np.exp(1j * 2 * np.pi * np.fft.rfftfreq(2 * (K - 1))))
    turns, logr = ops.zprep(z.to(DEV))
    delays = torch.tensor(np.random.RandomState(3).randint(600, 1600, nblk * n).astype(np.float32)).to(DEV)
    Q, QQ, coef, coef_sub = ops.tf_ortho_coefs(M, ig, b, c)
    scale = (0.5 + torch.rand(nblk, generator=gen)).to(DEV)
    eye = torch.eye(G, device=DEV).repeat(nb, 1).contiguous()
    filt = torch.view_as_complex(torch.randn(nb, K, 2, generator=gen).to(DEV)).contiguous()
    _, Ts_s = ops.tf_compose_fwd(turns, None, coef, delays, n, eye, scale, None, filt, None, nb, save_T=True)
    _, Ts_u = ops.tf_compose_fwd(turns, None, coef, delays, n, eye, None, None, filt, None, nb, save_T=True)
    assert rel_err(Ts_s.cpu(), (Ts_u * scale[:, None]).cpu()) < 1e-6
    gH = torch.view_as_complex(torch.randn(nblk, K, 2, generator=gen).to(DEV)).contiguous()
    ref = ops.tf_compose_bwd(turns, None, coef, delays, n, eye, gH, Ts_s, filt, nb)
    got = ops.tf_compose_bwd(turns, None, coef, delays, n, eye, gH, Ts_u, filt, nb, tscale=scale)
    assert rel_err(got.cpu(), ref.cpu()) < 2e-6


@pytest.mark.parametrize("n", [4, 3, 2])
def test_fused_tail_equals_separate_launches(ops, n):
    """gfdn_tf_tail == gfdn_tf_param_grads, then Adam on the flat buffers, then gfdn_tf_ortho_coefs of the updated blocks:
    gradients, parameters, both moments, the step counter and the next step's Q / QQ / record sets, two steps in a row (the
    second one reads the first one's in-place results)."""
    from diffgfdn_amd.optim import FlatAdam
    gen = torch.Generator(device="cpu").manual_seed(40 + n)
    nblk, parts = 6, 9
    M0, b0, c0, ig = _bank_blocks(gen, nblk, n)

    def build():
        c = torch.nn.Parameter(c0.clone())
        b = torch.nn.Parameter(b0.clone())
        M = torch.nn.Parameter(M0.clone())
        w = torch.nn.Parameter(torch.zeros(5, device=DEV))
        opt = FlatAdam([{'params': [c], 'lr': 1e-2}, {'params': [b], 'lr': 1e-2}, {'params': [M], 'lr': 1e-3},
                        {'params': [w], 'lr': 1e-2}])
        return opt, M, b, c

    opt_a, Ma, ba, ca = build()
    opt_b, Mb, bb, cb = build()
    rec_b = tuple(torch.empty_like(t) for t in ops.tf_ortho_coefs(Mb.data, ig, bb.data, cb.data))
    ops.tf_ortho_coefs(Mb.data, ig, bb.data, cb.data, out=rec_b)
    offs = (opt_b.flat_range(Mb)[0], opt_b.flat_range(bb)[0], opt_b.flat_range(cb)[0])
    views = {id(p): v for p, v in zip(opt_b._params, opt_b._grad_views)}
    for step in range(2):
        grec0 = torch.randn(nblk, 32, parts, generator=gen).to(DEV)
        grec1 = torch.randn(nblk, 32, generator=gen).to(DEV)
        gQ = torch.randn(nblk, n, n, generator=gen).to(DEV)
        # separate launches
        Qa, QQa, _, _ = ops.tf_ortho_coefs(Ma.data, ig, ba.data, ca.data)
        va = {id(p): v for p, v in zip(opt_a._params, opt_a._grad_views)}
        ops.tf_param_grads(QQa, ig, grec0, ba.data, ca.data, Ma.data, A1=Ma.data, grec1=grec1, gQ=gQ, Q=Qa,
                           gb=va[id(ba)].view(-1), gc=va[id(ca)].view(-1), gM=va[id(Ma)])
        opt_a._packed = True
        opt_a.step()
        rec_a = ops.tf_ortho_coefs(Ma.data, ig, ba.data, ca.data)
        # the fused launch (the gradient of the fourth leaf is zero: its range is stepped by the plain kernel)
        Qb, QQb = rec_b[0], rec_b[1]
        ops.tf_tail(QQb, ig, grec0, grec1, bb.data, cb.data, Mb.data.view(nblk, n, n), gQ, Qb, views[id(bb)].view(-1),
                    views[id(cb)].view(-1), views[id(Mb)].view(-1), opt_b, *offs, *rec_b)
        opt_b.step_range(opt_b.flat_range(opt_b._params[3])[0], opt_b.flat_grad.numel(), second=True)
        torch.cuda.synchronize()
        assert torch.equal(opt_a.flat_grad, opt_b.flat_grad), step
        assert torch.equal(opt_a.flat_param, opt_b.flat_param), step
        assert torch.equal(opt_a.exp_avg, opt_b.exp_avg) and torch.equal(opt_a.exp_avg_sq, opt_b.exp_avg_sq), step
        assert float(opt_b.step_count) == step + 1 and float(opt_b.step_count2) == step + 1
        for got, want in zip(rec_b, rec_a):
            assert torch.equal(got, want), step


def test_merge_into_slot_order(ops):
    """gfdn_lin_merge_slots: a + b + c of pair-interleaved rows written in the adjoint transform's slot order == the sum
    scattered by torch; one and two inputs; time order without the table."""
    gen = torch.Generator(device="cpu").manual_seed(8)
    n, rows = 65537, 5
    a, b, c = (torch.randn(rows, n, 2, generator=gen).to(DEV) for _ in range(3))
    sot = ops.irfft_slot_of_time(n, torch.device(DEV))
    pos = torch.cat([torch.zeros(1, dtype=torch.long, device=DEV), 1 + sot[1:].long()])
    for parts in ((a,), (a, b), (a, b, c)):
        want = torch.empty_like(a)
        tot = parts[0]
        for p in parts[1:]:
            tot = tot + p
        want[:, pos] = tot
        got = ops.lin_merge_slots(*parts, slot_of_time=sot)
        assert torch.equal(got, want)
        assert torch.equal(ops.lin_merge_slots(*parts), tot)


def test_gain_network_backward_with_the_scale_in_the_gains(ops):
    """mlp_gains_bwd(ggains_parts, colscale) on rows of dL/d(gains s) == the backward on the rows times s
    (bankstep.FusedBankStep.scale_in_gains)."""
    from diffgfdn_amd import _lib
    nb, Bper, G, F, H, nh, chunks = 3, 8, 4, 20, 16, 2, 37
    B = nb * Bper
    g = torch.Generator().manual_seed(21)
    pos = torch.rand(B, 3, generator=g, dtype=torch.float64).to(DEV)
    P = _lib.load().gfdn_mlp_param_count(F, H, nh, G)
    w = (0.3 * torch.randn(nb, P, generator=g)).to(DEV)
    freq_pi = (torch.exp(torch.linspace(0.0, np.log(32.0), F)) * np.pi).float().to(DEV)
    s = (0.5 + torch.rand(nb * G, generator=g)).to(DEV)
    assert ops.mlp_bwd_takes_parts(F, H, nh, G, Bper)
    gains, xhat, rstd = ops.mlp_gains_fwd(pos, freq_pi, w, H, nh, G, 0.0, 1.0, None, nb)
    srow = s.reshape(nb, 1, G).expand(nb, Bper, G).reshape(B, G)
    parts = torch.randn(B * G, chunks, generator=g).to(DEV)
    want = ops.mlp_gains_bwd(pos, freq_pi, w, H, nh, G, 0.0, 1.0, gains, xhat, rstd, None, None, nb,
                             ggains_parts=(parts * srow.reshape(-1, 1)).contiguous())
    got = ops.mlp_gains_bwd(pos, freq_pi, w, H, nh, G, 0.0, 1.0, gains, xhat, rstd, None, None, nb, ggains_parts=parts,
                            colscale=s)
    assert rel_err(got.cpu(), want.cpu()) < 2e-6


@pytest.mark.parametrize("n", [4, 3])
def test_energy_finish_stores_the_scaled_gains(ops, n):
    """tf_energy(gains=): same energy / scale / rescaled b, c as without, and gains * scale per (band, group) column, exactly."""
    gen = torch.Generator(device="cpu").manual_seed(31 + n)
    nb, G, K, Bper = 2, 4, 4097, 6
    nblk = nb * G
    M, b, c, ig = _bank_blocks(gen, nblk, n)
    z = torch.tensor(np.exp(1j * 2 * np.pi * np.fft.rfftfreq(2 * (K - 1))))
    turns, logr = ops.zprep(z.to(DEV))
    delays = torch.tensor(np.random.RandomState(9).randint(600, 1600, nblk * n).astype(np.float32)).to(DEV)
    Q, QQ, coef, coef_sub = ops.tf_ortho_coefs(M, ig, b, c)
    gains = torch.rand(nb * Bper, G, generator=gen).to(DEV)
    b1, c1, b2, c2 = b.clone(), c.clone(), b.clone(), c.clone()
    e1, s1 = ops.tf_energy(turns, None, coef_sub, delays, n, b1, c1, dturn=0.5 / (K - 1))
    e2, s2, gs = ops.tf_energy(turns, None, coef_sub, delays, n, b2, c2, dturn=0.5 / (K - 1), gains=gains, G=G)
    assert torch.equal(e1, e2) and torch.equal(s1, s2) and torch.equal(b1, b2) and torch.equal(c1, c2)
    srow = s1.reshape(nb, 1, G).expand(nb, Bper, G).reshape(nb * Bper, G)
    assert torch.equal(gs, gains * srow)


@pytest.mark.parametrize("n", [4, 3])
def test_records_pass_with_the_scale_in_the_gains(ops, n):
    """gfdn_tf_compose_bwd(tscale, gain_fold) on s dL/dT' == the pass on dL/dT' (what the adjoint transform hands over when the
    group signals are those of the unscaled functions and the receiver gains carry the scale)."""
    gen = torch.Generator(device="cpu").manual_seed(17 + n)
    nb, G, K = 2, 4, 4097
    nblk = nb * G
    M, b, c, ig = _bank_blocks(gen, nblk, n)
    z = torch.tensor(np.exp(1j * 2 * np.pi * np.fft.rfftfreq(2 * (K - 1))))
    turns, logr = ops.zprep(z.to(DEV))
    delays = torch.tensor(np.random.RandomState(5).randint(600, 1600, nblk * n).astype(np.float32)).to(DEV)
    Q, QQ, coef, coef_sub = ops.tf_ortho_coefs(M, ig, b, c)
    scale = (0.5 + torch.rand(nblk, generator=gen)).to(DEV)
    eye = torch.eye(G, device=DEV).repeat(nb, 1).contiguous()
    filt = torch.view_as_complex(torch.randn(nb, K, 2, generator=gen).to(DEV)).contiguous()
    _, Ts_u = ops.tf_compose_fwd(turns, None, coef, delays, n, eye, None, None, filt, None, nb, save_T=True)
    gH = torch.view_as_complex(torch.randn(nblk, K, 2, generator=gen).to(DEV)).contiguous()
    ref = ops.tf_compose_bwd(turns, None, coef, delays, n, eye, gH, Ts_u, filt, nb, tscale=scale)
    got = ops.tf_compose_bwd(turns, None, coef, delays, n, eye, (gH * scale[:, None]).contiguous(), Ts_u, filt, nb, tscale=scale,
                             gain_fold=True)
    assert rel_err(got.cpu(), ref.cpu()) < 2e-6
