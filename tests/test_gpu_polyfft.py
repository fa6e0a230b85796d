"""csrc/polyfft.hip: the 8-line block transfer functions on the rfftfreq grid by fast transforms -- against a float64
evaluation of the same polynomials and against the matrix-core passes of csrc/blocktf8.hip (gfdn_tf8_*), which stay the
general path (any grid, any delay lengths)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(nfft, nbands=2, G=2, nper=8, seed=0, dmax=None):
    from diffgfdn_amd import hip_ops as ops
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(seed)
    nblk = nbands * G
    # (raw blocks of spectral radius ~0.6 and 1 / gamma of a decaying loop: poles away from the unit circle, where float32
    # evaluations of 1 / Q agree to ~1e-5 -- next to a pole they differ by the pole's condition, whichever way Q is formed)
    M = ((2 * torch.rand(nblk, nper, nper, generator=g) - 1) / np.sqrt(nper)).to(dev)
    b = (0.3 + torch.rand(nblk * nper, generator=g)).to(dev)
    c = (0.3 + torch.rand(nblk * nper, generator=g)).to(dev)
    ig = (1.05 + 0.25 * torch.rand(nblk * nper, generator=g)).to(dev)
    dmax = dmax or max(nfft // 6, 40)
    delays = torch.randint(7, dmax, (nblk * nper,), generator=g).to(torch.float32).to(dev)
    Q, QQ = ops.ortho_fwd(M, True, True)
    coef, coef_sub = ops.tf8_coefs(QQ, ig, b, c, A1=M)
    K = nfft // 2 + 1
    turns = (torch.arange(K, dtype=torch.float64) / nfft).to(dev)
    return dict(ops=ops, dev=dev, nblk=nblk, nbands=nbands, G=G, nper=nper, M=M, b=b, c=c, ig=ig, delays=delays, QQ=QQ,
                coef=coef, coef_sub=coef_sub, K=K, nfft=nfft, turns=turns, dturn=1.0 / nfft)


def _poly64(coef, delays, c, nper, turns):
    """(Q, P) (nblk, K) complex128 of the records at the grid, subset by subset"""
    coef = coef.double().cpu().numpy()
    d = delays.double().cpu().numpy().reshape(-1, nper)
    cc = c.double().cpu().numpy().reshape(-1, nper)
    t = turns.cpu().numpy()
    nblk = coef.shape[0]
    Qv = np.zeros((nblk, t.size), dtype=np.complex128)
    Pv = np.zeros_like(Qv)
    for blk in range(nblk):
        for S in range(256):
            if S >> nper:
                continue
            m = sum(d[blk, i] for i in range(nper) if (S >> i) & 1)
            e = np.exp(2j * np.pi * ((m * t) % 1.0))
            Qv[blk] += coef[blk, 0, S] * e
            Pv[blk] += sum(cc[blk, i] * coef[blk, 1 + i, S] for i in range(nper)) * e
    return Qv, Pv


@pytest.mark.parametrize("nfft,nper", [(2048, 8), (512, 6), (131072, 8)])
def test_transformed_sequences_equal_direct_evaluation(nfft, nper):
    s = _setup(nfft, nper=nper, seed=nfft % 97, dmax=2400 if nfft > 4096 else None)
    ops = s['ops']
    T = ops.tfp_plan(s['delays'], nper, nfft)
    assert T is not None and T % 256 == 0 and T <= nfft
    nblk = s['nblk']
    for set_, coef in ((0, s['coef']), (1, s['coef_sub'])):
        X = ops.tfp_forward(coef, s['delays'], s['c'], nper, nfft, T)
        assert tuple(X.shape) == (2 * nblk, s['K'])
        sel = torch.arange(0, s['K'], 1 if nfft <= 4096 else 61, device=X.device)      # (a sample of the bins at full size)
        sel = torch.unique(torch.cat([sel, torch.tensor([0, s['K'] - 1, s['K'] // 2], device=X.device)]))
        Qv, Pv = _poly64(coef, s['delays'], s['c'], nper, s['turns'][sel])
        got_q = X[:nblk][:, sel].conj().cpu().numpy()
        got_p = X[nblk:][:, sel].conj().cpu().numpy()
        for got, ref in ((got_q, Qv), (got_p, Pv)):
            scale = np.abs(ref).max(axis=1, keepdims=True)
            assert np.abs(got - ref).max() / scale.max() < 3e-6
            assert (np.abs(got - ref) / scale).max() < 3e-6


def test_non_integer_delays_have_no_plan():
    s = _setup(2048)
    d = s['delays'].clone()
    d[3] += 0.5
    assert s['ops'].tfp_plan(d, 8, 2048) is None


@pytest.mark.parametrize("nfft,nper", [(2048, 8), (2048, 5), (131072, 8)])
def test_energy_and_colorless_equal_matrix_core_passes(nfft, nper):
    s = _setup(nfft, nper=nper, seed=5, dmax=2400 if nfft > 4096 else None)
    ops, nblk = s['ops'], s['nblk']
    T = ops.tfp_plan(s['delays'], nper, nfft)
    X = ops.tfp_forward(s['coef_sub'], s['delays'], s['c'], nper, nfft, T)
    Xq, Xp = X[:nblk], X[nblk:]
    b1, c1 = s['b'].clone(), s['c'].clone()
    b2, c2 = s['b'].clone(), s['c'].clone()
    e_ref, sc_ref = ops.tf8_energy(s['turns'], s['coef_sub'], s['delays'], nper, b1, c1, want_energy=True, dturn=s['dturn'])
    e_got, sc_got = ops.tfp_energy(Xq, Xp, nper, b2, c2, want_energy=True)
    # (the raw sub-FDN blocks are lossless: their poles sit ON the unit circle and a few bins next to them carry the energy
    # -- float32 roundings of Q there move E by ~1e-4 in either evaluation; the float64 value decides)
    if nfft <= 4096:
        Qv, Pv = _poly64(s['coef_sub'], s['delays'], s['c'], nper, s['turns'])
        e64 = torch.tensor((np.abs(Pv / Qv) ** 2).mean(axis=1), device=e_got.device)
        assert ((e_got.double() - e64).abs() / e64).max() < 3e-4
        assert ((e_got.double() - e64).abs() / e64).max() < 3 * ((e_ref.double() - e64).abs() / e64).max() + 3e-5
    gains = torch.rand(s['nbands'] * 5, s['G'], device=e_got.device)
    b3, c3 = s['b'].clone(), s['c'].clone()
    e3, sc3, gsc = ops.tfp_energy(Xq, Xp, nper, b3, c3, want_energy=True, gains=gains, G=s['G'])
    assert torch.equal(e3, e_got) and torch.equal(sc3, sc_got) and torch.equal(b3, b2)
    assert torch.equal(gsc, gains * sc_got.reshape(s['nbands'], 1, s['G']).expand(-1, 5, -1).reshape(-1, s['G']))
    torch.testing.assert_close(e_got, e_ref, rtol=3e-4, atol=0)
    torch.testing.assert_close(sc_got, sc_ref, rtol=2e-4, atol=0)
    torch.testing.assert_close(b2, b1, rtol=1e-4, atol=0)
    torch.testing.assert_close(c2, c1, rtol=1e-4, atol=0)
    for asym in (False, True):
        part_ref, loss_ref = ops.tf8_colorless(s['turns'], s['coef_sub'], s['delays'], nper, c1, sc_ref, asym, 0.7,
                                               dturn=s['dturn'])
        # (asym: the gather's T samples only; symmetric: the whole signals)
        part_got, loss_got = ops.tfp_colorless(Xq, Xp, nfft, nper, s['delays'], sc_ref, asym, 0.7, T=T if asym else None)
        torch.testing.assert_close(loss_got, loss_ref, rtol=3e-4, atol=1e-7)
        ref = part_ref.sum(-1)
        got = part_got.sum(-1)
        present = torch.tensor([(S >> nper) == 0 for S in range(256)] * 2, device=ref.device)
        tol = 5e-4 * ref[:, present].abs().max()
        assert (got[:, present] - ref[:, present]).abs().max() < tol, ((got - ref)[:, present].abs().max(), tol)
        assert got[:, ~present].abs().max() == 0 if (~present).any() else True


@pytest.mark.parametrize("nfft,nper", [(2048, 8), (131072, 8)])
def test_output_stage_in_slot_order_and_adjoint_by_transforms(nfft, nper):
    s = _setup(nfft, nper=nper, seed=11, dmax=2400 if nfft > 4096 else None)
    ops, nblk, nbands, G, dev = s['ops'], s['nblk'], s['nbands'], s['G'], s['dev']
    K = s['K']
    Ku = K // 2 + 1
    g = torch.Generator().manual_seed(3)
    # a slot order of our own: column 0 = bin 0, the others a permutation of bins 1 .. Ku - 1, some conjugated
    perm = 1 + torch.randperm(Ku - 1, generator=g)
    conj = torch.rand(Ku - 1, generator=g) < 0.5
    col = torch.zeros(Ku, dtype=torch.int64)
    col[perm] = torch.arange(1, Ku, dtype=torch.int64) | (conj.to(torch.int64) << 31)
    col = torch.where(col >= 2 ** 31, col - 2 ** 32, col).to(torch.int32).to(dev)
    turns_nat = s['turns'][:Ku]
    tslot = torch.cat([turns_nat[:1], torch.where(conj.to(dev), -turns_nat[perm.to(dev)], turns_nat[perm.to(dev)])])
    filt = torch.complex(torch.randn(nbands, Ku, generator=g), torch.randn(nbands, Ku, generator=g)).to(torch.complex64).to(dev)
    scale = (0.5 + torch.rand(nblk, generator=g)).to(dev)
    # (gains "after the rescale" for the passes that take the scale themselves: c' = c sqrt(scale))
    c_new = (s['c'].reshape(nblk, nper) * scale.sqrt()[:, None]).reshape(-1).contiguous()
    # the forward pass on the grid in BIN order, unscaled, its group responses scattered to the slot order ...
    Tn, _, Hg, Dn = ops.tf8_tsave(turns_nat.contiguous(), s['coef'], s['delays'], nper, s['c'], None, nbands, G, quad=False,
                                  filt=filt, want_H=True, hslot=col)
    # ... against the pass on the slot-ordered grid itself (scaled)
    Ts_ref, _, Hg_ref, Dinv_ref = ops.tf8_tsave(tslot, s['coef'], s['delays'], nper, c_new, scale, nbands, G, quad=False,
                                                filt=filt, want_H=True)
    Hg_scaled = Hg * scale[:, None]
    assert (Hg_scaled - Hg_ref).abs().max() < 2e-5 * Hg_ref.abs().max(), (Hg_scaled - Hg_ref).abs().max()
    colx = (col & 0x7fffffff).long()
    cj = (col < 0)[None, :]
    Ts_slot_of_bin = Ts_ref[:, colx]
    assert (torch.where(cj, Ts_slot_of_bin.conj(), Ts_slot_of_bin) - Tn * scale[:, None]).abs().max() < 2e-5 * Ts_ref.abs().max()
    Di = Dinv_ref[:, colx]
    assert (torch.where(cj, Di.conj(), Di) - Dn).abs().max() < 2e-5 * Dinv_ref.abs().max()
    # adjoint by inverse transforms: one gradient row per group (identity gains), T' = scale Tn
    gH = torch.complex(torch.randn(nblk, Ku, generator=g), torch.randn(nblk, Ku, generator=g)).to(torch.complex64).to(dev)
    eye = torch.eye(G, device=dev).repeat(nbands, 1)
    part_ref = ops.tf8_compose_bwd(tslot, s['coef'], s['delays'], nper, c_new, scale, eye, gH, filt, nbands,
                                   saved=(Ts_ref, Dinv_ref)).sum(-1)
    part_got = ops.tfp_compose_bwd(nfft, nbands, G, nper, s['delays'], Ku, col, gH, filt, Tn, Dn, tscale=scale,
                                   T=ops.tfp_plan(s['delays'], nper, nfft)).sum(-1)
    tol = 2e-4 * part_ref.abs().max()
    assert (part_got - part_ref).abs().max() < tol, ((part_got - part_ref).abs().max(), tol)
    # the scale in the receiver gains: s dL/dT' comes in
    part_fold = ops.tfp_compose_bwd(nfft, nbands, G, nper, s['delays'], Ku, col, (gH * scale[:, None]).contiguous(), filt, Tn, Dn,
                                    tscale=scale, gain_fold=True).sum(-1)
    assert (part_fold - part_got).abs().max() < 2e-5 * part_got.abs().max()
