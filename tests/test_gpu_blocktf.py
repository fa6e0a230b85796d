"""Block transfer functions in polynomial form (csrc/blocktf.hip) against a float64 / complex128 torch
restatement of the same quantities from the per-bin SOLVE (feedback_loop.py:326-391, model.py:209-252,
:583-619): records + evaluation, normalisation energy, colorless loss + gradients, output stage forward and
backward, and the record -> (A, b, c) map."""
import numpy as np
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _grid(K, radius=1.0, seed=0):
    k = torch.arange(K, dtype=torch.float64)
    z = torch.polar(torch.full((K,), radius, dtype=torch.float64), np.pi * k / (K - 1))
    return z


def _blocks(nblk, n, seed, orth=False):
    g = torch.Generator().manual_seed(seed)
    A = (2 * torch.rand(nblk, n, n, generator=g, dtype=torch.float64) - 1) / np.sqrt(n)
    if orth:
        X = torch.triu(A, 1)
        Q = torch.linalg.matrix_exp(X - X.transpose(1, 2))
        A = Q @ Q
    b = (2 * torch.randn(nblk * n, generator=g, dtype=torch.float64) - 1) / (nblk * n) * 3
    c = (2 * torch.randn(nblk * n, generator=g, dtype=torch.float64) - 1) / (nblk * n) * 3
    primes = [641, 653, 701, 809, 977, 1031, 1201, 1301, 1423, 1511, 1583, 1601, 659, 743, 887, 1093]
    delays = torch.tensor([primes[(7 * i + seed) % len(primes)] + 2 * (i // len(primes)) for i in range(nblk * n)],
                          dtype=torch.float64)
    ig = 1.0 / 10 ** (-3 * delays / (32000.0 * (0.3 + 1.2 * torch.rand(nblk * n, generator=g, dtype=torch.float64))))
    return A, b, c, delays, ig


def _T_ref(z, A, b, c, delays, ig):
    """T (K, nblk) complex128 by the per-bin solve."""
    nblk, n, _ = A.shape
    out = []
    for q in range(nblk):
        sl = slice(q * n, (q + 1) * n)
        D = torch.diag_embed(z[:, None] ** delays[sl][None, :] * ig[sl][None, :])
        Pm = D - A[q].to(torch.complex128)[None]
        y = torch.linalg.solve(Pm, b[sl].to(torch.complex128)[None, :, None].expand(z.numel(), n, 1))
        out.append((c[sl].to(torch.complex128)[None, :] * y[..., 0]).sum(-1))
    return torch.stack(out, dim=1)


@pytest.mark.parametrize("n,nblk,radius,orth", [(4, 6, 1.0, True), (4, 28, 1.0, False), (3, 5, 1.0, True),
                                                 (2, 3, 1.0, False), (4, 4, 1.0002, True), (1, 2, 1.0, False)])
def test_records_and_evaluation(n, nblk, radius, orth):
    from diffgfdn_amd import hip_ops as ops
    K = 2049
    z = _grid(K, radius)
    A, b, c, delays, ig = _blocks(nblk, n, 3, orth)
    if not orth:
        ig = torch.ones_like(ig)
    turns, logr = ops.zprep(z.to(DEV))
    coef = ops.tf_coefs(A.to(DEV), b.to(DEV), c.to(DEV), None if not orth else ig.to(DEV))
    T = ops.tf_eval(turns, logr if radius != 1.0 else None, coef, delays.to(DEV), n)
    ref = _T_ref(z, A.float().double(), b.float().double(), c.float().double(), delays, ig.float().double())
    assert rel_err(T.cpu().numpy(), ref.numpy()) < 2e-5
    c0, c1 = ops.tf_coefs2(A.to(DEV), None if not orth else ig.to(DEV), A.transpose(1, 2).contiguous().to(DEV), None,
                           b.to(DEV), c.to(DEV))
    assert torch.equal(c0, coef)
    assert torch.equal(c1, ops.tf_coefs(A.transpose(1, 2).contiguous().to(DEV), b.to(DEV), c.to(DEV)))
    # energy + scale + in-place rescale (trainer.py:317-332)
    bb, cc = b.float().to(DEV).contiguous(), c.float().to(DEV).contiguous()
    energy, scale = ops.tf_energy(turns, logr if radius != 1.0 else None, coef, delays.to(DEV), n, bb, cc)
    E = (ref.abs() ** 2).mean(0)
    assert rel_err(energy.cpu().numpy(), E.numpy()) < 2e-5
    if radius == 1.0:          # uniform grid on the unit circle: the pass that steps the phasors by constant rotations
        e2, _ = ops.tf_energy(turns, None, coef, delays.to(DEV), n, dturn=0.5 / (K - 1))
        assert rel_err(e2.cpu().numpy(), E.numpy()) < 2e-5
    assert rel_err(scale.cpu().numpy(), (E ** -0.5).numpy()) < 2e-5
    d = (E ** 0.25).repeat_interleave(n)
    assert rel_err(bb.cpu().numpy(), (b.float().double() / d).numpy()) < 1e-5
    assert rel_err(cc.cpu().numpy(), (c.float().double() / d).numpy()) < 1e-5


def _colorless_ref(z, M, b, c, delays, s, asym, gscale):
    ones = torch.ones_like(delays)
    S = _T_ref(z, M, b, c, delays, ones) * s[None, :]
    d = S.abs() - 1
    per = torch.where((d > 1) & asym, d ** 4, d ** 2) if asym else d ** 2
    loss_g = per.mean(0)
    return gscale * loss_g.sum(), loss_g


@pytest.mark.parametrize("runs", [False, True])
@pytest.mark.parametrize("n,nblk,asym", [(4, 28, True), (4, 6, False), (3, 4, True)])
def test_colorless_pass_and_record_adjoint(n, nblk, asym, runs):
    """loss_g and d(gscale sum_g loss_g)/d(M, b, c) at the scaled responses against float64 autograd."""
    from diffgfdn_amd import hip_ops as ops
    K = 4097
    z = _grid(K)
    M, b, c, delays, _ = _blocks(nblk, n, 5)
    b, c = b * 4, c * 4                                  # responses around 1 and beyond: both branches of amse
    M, b, c = M.float().double(), b.float().double(), c.float().double()
    s = (0.5 + torch.rand(nblk, dtype=torch.float64)).float().double()
    gscale = 0.7
    turns, _ = ops.zprep(z.to(DEV))
    # records of the SCALED responses are taken at b' c'^T = s b c^T: b' = sqrt(s) b, c' = sqrt(s) c
    rs = s.sqrt().repeat_interleave(n)
    bp, cp = (b * rs).requires_grad_(), (c * rs).requires_grad_()
    Mr = M.clone().requires_grad_()
    L, loss_g = _colorless_ref(z, Mr, bp, cp, delays, torch.ones(nblk, dtype=torch.float64), asym, gscale)
    L.backward()
    coef = ops.tf_coefs(M.to(DEV), b.to(DEV), c.to(DEV))
    grec, loss = ops.tf_colorless(turns, None, coef, delays.to(DEV), n, s.to(DEV), asym, gscale,
                                  dturn=0.5 / (K - 1) if runs else 0.0)
    assert rel_err(loss.cpu().numpy(), loss_g.detach().numpy()) < 2e-5
    assert torch.equal(grec[:, 15], loss)
    # one record set through the two-set entry point: set 0 a dummy with zero gradient records
    gA0, gM, gb, gc = ops.tf_coefs_bwd(M.to(DEV), None, torch.zeros_like(grec), bp.detach().float().to(DEV),
                                       cp.detach().float().to(DEV), A1=M.to(DEV), grec1=grec)
    assert float(gA0.abs().max()) == 0.0
    assert rel_err(gM.cpu().numpy(), Mr.grad.numpy()) < 1e-4
    assert rel_err(gb.cpu().numpy(), bp.grad.numpy()) < 1e-4
    assert rel_err(gc.cpu().numpy(), cp.grad.numpy()) < 1e-4


@pytest.mark.parametrize("n,G,nbands,B,K", [(4, 4, 3, 8, 1500), (4, 3, 1, 5, 777), (3, 4, 2, 32, 1025), (4, 4, 7, 32, 4100),
                                             (4, 2, 2, 13, 3000)])
def test_output_stage_from_records(n, G, nbands, B, K):
    """H = (sum_g rgain s_g T_g + direct[rows]) filt and the adjoint: grgain, gQQ, gb, gc against autograd."""
    from diffgfdn_amd import hip_ops as ops
    nblk = nbands * G
    z = _grid(K)
    A, b, c, delays, ig = _blocks(nblk, n, 11, orth=True)
    A, b, c, ig = A.float().double(), b.float().double(), c.float().double(), ig.float().double()
    g = torch.Generator().manual_seed(1)
    s = (0.5 + torch.rand(nblk, generator=g, dtype=torch.float64)).float().double()
    rgain = (2 * torch.rand(nbands * B, G, generator=g, dtype=torch.float64) - 1).float().double()
    R = 3 * B
    direct = torch.randn(nbands * R, K, generator=g, dtype=torch.complex128).to(torch.complex64)
    rows = torch.stack([q * R + torch.randperm(R, generator=g)[:B] for q in range(nbands)]).reshape(-1)
    filt = torch.randn(nbands, K, generator=g, dtype=torch.complex128).to(torch.complex64)
    W = torch.randn(nbands * B, K, generator=g, dtype=torch.complex128).to(torch.complex64)     # dL/dH

    rs = s.sqrt().repeat_interleave(n)
    bp, cp = (b * rs).requires_grad_(), (c * rs).requires_grad_()
    Ar, rg = A.clone().requires_grad_(), rgain.clone().requires_grad_()
    T = _T_ref(z, Ar, bp, cp, delays, ig)                                           # scaled responses (K, nblk)
    Tb = T.reshape(K, nbands, G).permute(1, 2, 0)                                    # (nbands, G, K)
    H = (torch.einsum('qbg,qgk->qbk', rg.reshape(nbands, B, G).to(torch.complex128), Tb)
         + direct[rows].to(torch.complex128).reshape(nbands, B, K)) * filt.to(torch.complex128)[:, None, :]
    H = H.reshape(nbands * B, K)
    L = (H.real * W.real.double() + H.imag * W.imag.double()).sum()
    L.backward()

    turns, _ = ops.zprep(z.to(DEV))
    coef = ops.tf_coefs(A.to(DEV), b.to(DEV), c.to(DEV), ig.to(DEV))
    dl = delays.to(DEV)
    Hd, Ts = ops.tf_compose_fwd(turns, None, coef, dl, n, rgain.to(DEV), s.to(DEV), direct.to(DEV), filt.to(DEV),
                                rows.to(DEV), nbands, save_T=True)
    assert rel_err(Hd.cpu().numpy(), H.detach().numpy()) < 2e-5
    assert rel_err(Ts.cpu().numpy(), T.detach().T.numpy()) < 2e-5
    grg = ops.tf_gain_grad(Ts, W.to(DEV), G, filt.to(DEV), nbands)
    grec = ops.tf_compose_bwd(turns, None, coef, dl, n, rgain.to(DEV), W.to(DEV), Ts, filt.to(DEV), nbands)
    assert rel_err(grg.cpu().numpy(), rg.grad.numpy()) < 1e-4
    gA, _, gb, gc = ops.tf_coefs_bwd(A.to(DEV), ig.to(DEV), grec, bp.detach().float().to(DEV),
                                     cp.detach().float().to(DEV))
    assert rel_err(gA.cpu().numpy(), Ar.grad.numpy()) < 1e-4
    assert rel_err(gb.cpu().numpy(), bp.grad.numpy()) < 1e-4
    assert rel_err(gc.cpu().numpy(), cp.grad.numpy()) < 1e-4


def test_ortho_bwd_add():
    from diffgfdn_amd import hip_ops as ops
    torch.manual_seed(0)
    M = torch.randn(5, 4, 4, device=DEV) * 0.4
    gQ, gQQ, add = torch.randn_like(M), torch.randn_like(M), torch.randn_like(M)
    Q, _ = ops.ortho_fwd(M, True, False)
    base = ops.ortho_bwd(M, gQ, gQQ, Q)
    out = ops.ortho_bwd_add(M, gQ, gQQ, Q, add)
    assert torch.allclose(out, base + add, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("n,G,nbands,B,K,two_sets", [(4, 4, 7, 32, 4100, True), (3, 4, 2, 8, 1025, True),
                                                      (4, 2, 1, 5, 40000, False), (2, 3, 2, 4, 600, True)])
def test_param_grads_equals_separate_launches(n, G, nbands, B, K, two_sets):
    """gfdn_tf_param_grads (partial-row sums + records -> dL/dA, dL/db, dL/dc + expm adjoint in one launch) against
    compose_bwd's own row sum -> tf_coefs_bwd -> ortho_bwd_add: bit for bit."""
    from diffgfdn_amd import hip_ops as ops
    nblk = nbands * G
    g = torch.Generator().manual_seed(n * 100 + G)
    M = (0.4 * torch.randn(nblk, n, n, generator=g)).to(DEV)
    Q, QQ = ops.ortho_fwd(M, True, True)
    _, b, c, delays, ig = _blocks(nblk, n, 5, orth=True)
    b, c, ig, dl = b.float().to(DEV), c.float().to(DEV), ig.float().to(DEV), delays.to(DEV)
    z = _grid(K)
    turns, _ = ops.zprep(z.to(DEV))
    coef = ops.tf_coefs(QQ, b, c, ig)
    rgain = (2 * torch.rand(nbands * B, G, generator=g) - 1).to(DEV)
    s = (0.5 + torch.rand(nblk, generator=g)).to(DEV)
    W = torch.randn(nbands * B, K, generator=g, dtype=torch.complex64).to(DEV)
    R = 2 * B
    direct = torch.randn(nbands * R, K, generator=g, dtype=torch.complex64).to(DEV)
    rows = torch.stack([q * R + torch.randperm(R, generator=g)[:B] for q in range(nbands)]).reshape(-1).to(DEV)
    _, Ts = ops.tf_compose_fwd(turns, None, coef, dl, n, rgain, s, direct, None, rows, nbands, save_T=True)
    grec = ops.tf_compose_bwd(turns, None, coef, dl, n, rgain, W, Ts, None, nbands)
    parts = ops.tf_compose_bwd(turns, None, coef, dl, n, rgain, W, Ts, None, nbands, partial=True)
    assert parts.shape[:2] == (nblk, 32) and torch.equal(parts.sum(-1).isfinite(), torch.ones_like(grec, dtype=torch.bool))
    gQ = torch.randn(nblk, n, n, generator=g).to(DEV)
    grec1 = torch.randn(nblk, 32, generator=g).to(DEV) if two_sets else None
    gQQ, gMsub, gb, gc = ops.tf_coefs_bwd(QQ, ig, grec, b, c, A1=M if two_sets else None, grec1=grec1)
    want = ops.ortho_bwd_add(M, gQ, gQQ, Q, gMsub)
    for g0 in (parts, grec):                         # partial rows summed in the kernel / records already summed
        gM, gb2, gc2 = ops.tf_param_grads(QQ, ig, g0, b, c, M, A1=M if two_sets else None, grec1=grec1, gQ=gQ, Q=Q)
        assert torch.equal(gM, want) and torch.equal(gb2, gb) and torch.equal(gc2, gc)
    # without the forward's Q the kernel rebuilds it (float64): equal up to the rounding of the saved float32 Q
    gM3, _, _ = ops.tf_param_grads(QQ, ig, grec, b, c, M, A1=M if two_sets else None, grec1=grec1, gQ=gQ)
    assert torch.allclose(gM3, want, rtol=1e-4, atol=1e-5 * float(want.abs().max()))


@pytest.mark.parametrize("n,nblk", [(4, 28), (3, 5), (2, 4), (1, 3)])
def test_ortho_coefs_equals_separate_launches(n, nblk):
    """gfdn_tf_ortho_coefs (expm + both record sets in one launch) against ortho_fwd -> tf_coefs2: bit for bit."""
    from diffgfdn_amd import hip_ops as ops
    g = torch.Generator().manual_seed(n)
    M = (0.5 * torch.randn(nblk, n, n, generator=g)).to(DEV)
    _, b, c, _, ig = _blocks(nblk, n, 9, orth=True)
    b, c, ig = b.float().to(DEV), c.float().to(DEV), ig.float().to(DEV)
    Q0, QQ0 = ops.ortho_fwd(M, True, True)
    c0, c1 = ops.tf_coefs2(QQ0, ig, M, None, b, c)
    Q, QQ, coef, coef_sub = ops.tf_ortho_coefs(M, ig, b, c)
    assert torch.equal(Q, Q0) and torch.equal(QQ, QQ0) and torch.equal(coef, c0) and torch.equal(coef_sub, c1)
    _, _, coef_only, none = ops.tf_ortho_coefs(M, ig, b, c, sub=False)
    assert none is None and torch.equal(coef_only, c0)
    assert torch.equal(ops.tf_coefs(QQ0, b, c, ig), c0)


@pytest.mark.parametrize("nbands,B,G,with_filt", [(2, 4, 4, True), (1, 6, 3, False), (3, 2, 2, True)])
def test_output_stage_folded_into_the_transform(nbands, B, G, with_filt):
    """gfdn_irfft_odd_pairs_compose_fwd (H formed inside the first pass of the paired transform, never stored) against
    tf_compose_fwd -> irfft_odd_fwd(slots, pairs): the time signals agree to float32 rounding; Tsave without H equals Tsave."""
    from diffgfdn_amd import hip_ops as ops
    n, npts = 4, 65537
    half = (npts + 1) // 2
    nblk = nbands * G
    g = torch.Generator().manual_seed(nbands * 10 + B)
    A, b, c, delays, ig = _blocks(nblk, n, 3, orth=True)
    A, b, c, ig, dl = A.float().to(DEV), b.float().to(DEV), c.float().to(DEV), ig.float().to(DEV), delays.to(DEV)
    z = torch.exp(2j * np.pi * torch.rand(half, generator=g, dtype=torch.float64)).to(DEV)     # any grid: the test is algebraic
    turns, _ = ops.zprep(z)
    coef = ops.tf_coefs(A, b, c, ig)
    s = (0.5 + torch.rand(nblk, generator=g)).to(DEV)
    rgain = (2 * torch.rand(nbands * B, G, generator=g) - 1).to(DEV)
    R = 3 * B
    direct = (torch.randn(nbands * R, half, generator=g, dtype=torch.float32)
              + 1j * torch.randn(nbands * R, half, generator=g, dtype=torch.float32)).to(torch.complex64).to(DEV)
    rows = torch.stack([q * R + torch.randperm(R, generator=g)[:B] for q in range(nbands)]).reshape(-1).to(DEV)
    filt = None
    if with_filt:
        filt = (torch.randn(nbands, half, generator=g) + 1j * torch.randn(nbands, half, generator=g)).to(torch.complex64).to(DEV)
    H, Ts = ops.tf_compose_fwd(turns, None, coef, dl, n, rgain, s, direct, filt, rows, nbands, save_T=True)
    Tq, Ts2 = ops.tf_compose_fwd(turns, None, coef, dl, n, rgain, s, direct, filt, rows, nbands, save_T=True, want_H=False)
    assert torch.equal(Ts, Ts2) and tuple(Tq.shape) == (nbands, half, 4)
    for q in range(nbands):
        for gi in range(4):
            assert torch.equal(Tq[q, :, gi], Ts[q * G + gi] if gi < G else torch.zeros_like(Ts[0]))
    want = ops.irfft_odd_fwd(H, npts, slots=True, pairs=True)
    got, h0 = ops.irfft_odd_pairs_compose_fwd(direct, rows, Tq, rgain, filt, npts, nbands)
    # (the same operations in the same order; the compiler contracts multiply-adds differently in the two kernels)
    assert torch.allclose(h0, H[:, 0].real, rtol=1e-6, atol=1e-6)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) < 2e-6 * scale
    if nbands > 1 and B > 2:                      # an odd number of items per band: a pair would straddle two bands
        keep = torch.cat([torch.arange(q * B, q * B + B - 1) for q in range(nbands)]).to(DEV)
        with pytest.raises(RuntimeError):
            ops.irfft_odd_pairs_compose_fwd(direct, rows[keep], Tq, rgain[keep], filt, npts, nbands)


@pytest.mark.parametrize("nbands,B,G,with_filt", [(2, 4, 4, True), (1, 6, 3, False)])
def test_gains_pass_folded_into_the_adjoint_transform(nbands, B, G, with_filt):
    """gfdn_irfft_odd_pairs_gains_bwd: dL/dH equals gfdn_irfft_odd_pairs_bwd's bit for bit, and the row sums of its gains
    partials equal gfdn_tf_gain_grad on that dL/dH (other summation order: float32 rounding)."""
    from diffgfdn_amd import hip_ops as ops
    npts = 65537
    half = (npts + 1) // 2
    g = torch.Generator().manual_seed(5 * nbands + B)
    batch = nbands * B
    g2 = torch.randn(batch // 2, npts, 2, generator=g).to(DEV)
    Ts = (torch.randn(nbands * G, half, generator=g) + 1j * torch.randn(nbands * G, half, generator=g)).to(torch.complex64).to(DEV)
    Tq = torch.zeros(nbands, half, 4, dtype=torch.complex64, device=DEV)
    for q in range(nbands):
        for gi in range(G):
            Tq[q, :, gi] = Ts[q * G + gi]
    filt = None
    if with_filt:
        filt = (torch.randn(nbands, half, generator=g) + 1j * torch.randn(nbands, half, generator=g)).to(torch.complex64).to(DEV)
    gH0 = ops.irfft_odd_pairs_bwd(g2, npts, batch)
    gH, gpart = ops.irfft_odd_pairs_bwd(g2, npts, batch, gains=(Tq, filt, nbands, G))
    assert torch.equal(gH, gH0)
    want = ops.tf_gain_grad(Ts, gH0, G, filt, nbands)
    got = ops.tf_rows_sum(gpart)
    assert tuple(got.shape) == (batch, G)
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())
