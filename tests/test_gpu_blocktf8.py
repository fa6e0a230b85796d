"""Block transfer functions of 5..8-line blocks on the matrix cores (csrc/blocktf8.hip) against a float64 / complex128
torch restatement through per-bin solves (feedback_loop.py:326-391, model.py:209-252, :583-619, trainer.py:317-332,
colorless_fdn/losses.py:20-73): polynomial records + evaluation, normalisation, colorless loss + gradients, output-stage
adjoint, and the tail through the orthogonal parameterisation."""
import numpy as np
import pytest
from tests.margins import within
import torch

from tests.helpers import rel_err
from tests.test_gpu_blocktf import _T_ref, _blocks, _grid

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("n,nblk,orth,K", [(8, 4, True, 4097), (8, 28, False, 1500), (6, 3, True, 2049), (5, 2, False, 777)])
def test_records_transfer_functions_and_normalize(n, nblk, orth, K):
    from diffgfdn_amd import hip_ops as ops
    z = _grid(K)
    A, b, c, delays, ig = _blocks(nblk, n, 3, orth)
    if not orth:
        ig = torch.ones_like(ig)
    A, b, c, ig = A.float().double(), b.float().double(), c.float().double(), ig.float().double()
    turns, _ = ops.zprep(z.to(DEV))
    coef, _ = ops.tf8_coefs(A.to(DEV), ig.to(DEV) if orth else None, b.to(DEV), c.to(DEV))
    assert tuple(coef.shape) == (nblk, 9, 256)
    # the determinant polynomial at z = 1 (all phasors 1): sum of the coefficients = det(diag(1 / gamma) - A)
    want = torch.stack([torch.linalg.det(torch.diag(ig[q * n:(q + 1) * n]) - A[q]) for q in range(nblk)])
    assert rel_err(coef[:, 0].double().sum(-1).cpu().numpy(), want.numpy()) < 1e-5
    ref = _T_ref(z, A, b, c, delays, ig)                                  # (K, nblk)
    G = 4 if nblk % 4 == 0 else nblk
    nb = nblk // G
    Ts, Tq = ops.tf8_tsave(turns, coef, delays.to(DEV), n, c.to(DEV), None, nb, G, quad=G <= 4)
    assert rel_err(Ts.cpu().numpy(), ref.T.numpy()) < 5e-5
    if Tq is not None:
        for q in range(nb):
            for g in range(4):
                want_q = Ts[q * G + g] if g < G else torch.zeros_like(Ts[0])
                assert torch.equal(Tq[q, :, g], want_q)
    # normalize: energy, scale, in-place rescale; then the scaled functions from the rescaled gains
    bb, cc = b.float().to(DEV).contiguous(), c.float().to(DEV).contiguous()
    energy, scale = ops.tf8_energy(turns, coef, delays.to(DEV), n, bb, cc, want_energy=True)
    E = (ref.abs() ** 2).mean(0)
    assert rel_err(energy.cpu().numpy(), E.numpy()) < 5e-5
    assert rel_err(scale.cpu().numpy(), (E ** -0.5).numpy()) < 5e-5
    # uniform grid: phasors stepped by constant rotations between exact evaluations
    b2, c2 = b.float().to(DEV).contiguous(), c.float().to(DEV).contiguous()
    e2, _ = ops.tf8_energy(turns, coef, delays.to(DEV), n, b2, c2, want_energy=True, dturn=0.5 / (K - 1))
    assert rel_err(e2.cpu().numpy(), E.numpy()) < 5e-5
    d = (E ** 0.25).repeat_interleave(n)
    assert rel_err(cc.cpu().numpy(), (c / d).numpy()) < 1e-5 and rel_err(bb.cpu().numpy(), (b / d).numpy()) < 1e-5
    Ts2, _ = ops.tf8_tsave(turns, coef, delays.to(DEV), n, cc, scale, nb, G, quad=False)
    assert rel_err(Ts2.cpu().numpy(), (ref / E.sqrt()[None, :]).T.numpy()) < 5e-5


def _ortho(M):
    X = torch.triu(M, 1)
    Q = torch.linalg.matrix_exp(X - X.transpose(1, 2))
    return Q, Q @ Q


@pytest.mark.parametrize("runs", [False, True])
@pytest.mark.parametrize("n,nblk,asym", [(8, 8, True), (6, 3, False)])
def test_colorless_pass_and_gradients(n, nblk, asym, runs):
    """loss_g and d(gscale sum_g loss_g)/d(M, b, c) of the raw sub-FDN blocks against float64 autograd."""
    from diffgfdn_amd import hip_ops as ops
    K = 40001
    z = _grid(K)
    M, b, c, delays, _ = _blocks(nblk, n, 5)
    b, c = b * 6, c * 6
    M, b, c = M.float().double(), b.float().double(), c.float().double()
    s = (0.5 + torch.rand(nblk, dtype=torch.float64)).float().double()
    gscale = 0.7
    rs = s.sqrt().repeat_interleave(n)
    bp, cp = (b * rs).requires_grad_(), (c * rs).requires_grad_()
    Mr = M.clone().requires_grad_()
    S = _T_ref(z, Mr, bp, cp, delays, torch.ones_like(delays))
    dd = S.abs() - 1
    per = torch.where((dd > 1) & asym, dd ** 4, dd ** 2) if asym else dd ** 2
    loss_g = per.mean(0)
    (gscale * loss_g.sum()).backward()
    turns, _ = ops.zprep(z.to(DEV))
    coef, _ = ops.tf8_coefs(M.to(DEV), None, b.to(DEV), c.to(DEV))
    part, loss = ops.tf8_colorless(turns, coef, delays.to(DEV), n, cp.detach().float().to(DEV), s.to(DEV), asym, gscale,
                                   dturn=0.5 / (K - 1) if runs else 0.0)
    assert rel_err(loss.cpu().numpy(), loss_g.detach().numpy()) < 5e-5
    # tail with an empty first set: dL/dM_raw reaches M directly
    zero = torch.zeros_like(part)
    Mdev = M.float().to(DEV)
    gM, gb, gc = ops.tf8_param_grads(Mdev, None, zero, cp.detach().float().to(DEV) * 0 + bp.detach().float().to(DEV),
                                     cp.detach().float().to(DEV), Mdev, A1=Mdev, part1=part)
    assert rel_err(gM.cpu().numpy(), Mr.grad.numpy()) < 2e-4
    assert rel_err(gb.cpu().numpy(), bp.grad.numpy()) < 2e-4
    assert rel_err(gc.cpu().numpy(), cp.grad.numpy()) < 2e-4


@pytest.mark.parametrize("n,G,nbands,B,K", [(8, 4, 2, 8, 1500), (8, 4, 7, 32, 4100), (7, 3, 1, 5, 777)])
def test_output_stage_adjoint_through_the_parameterisation(n, G, nbands, B, K):
    """dL/dM (through Q = expm(skew M), A = Q Q), dL/db, dL/dc of L = Re<W, H>, H = (sum_g rgain s_g T_g) filt, against
    float64 autograd; the forward's T' feeds gfdn_tf_gain_grad unchanged."""
    from diffgfdn_amd import hip_ops as ops
    nblk = nbands * G
    z = _grid(K)
    _, b, c, delays, ig = _blocks(nblk, n, 11, orth=True)
    g = torch.Generator().manual_seed(1)
    M = (0.4 * torch.randn(nblk, n, n, generator=g)).float().double()
    b, c, ig = (b * 4).float().double(), (c * 4).float().double(), ig.float().double()
    s = (0.5 + torch.rand(nblk, generator=g, dtype=torch.float64)).float().double()
    rgain = (2 * torch.rand(nbands * B, G, generator=g, dtype=torch.float64) - 1).float().double()
    filt = torch.randn(nbands, K, generator=g, dtype=torch.complex128).to(torch.complex64)
    W = torch.randn(nbands * B, K, generator=g, dtype=torch.complex128).to(torch.complex64)
    rs = s.sqrt().repeat_interleave(n)
    bp, cp = (b * rs).requires_grad_(), (c * rs).requires_grad_()
    Mr, rg = M.clone().requires_grad_(), rgain.clone().requires_grad_()
    _, QQr = _ortho(Mr)
    T = _T_ref(z, QQr, bp, cp, delays, ig)
    Tb = T.reshape(K, nbands, G).permute(1, 2, 0)
    H = torch.einsum('qbg,qgk->qbk', rg.reshape(nbands, B, G).to(torch.complex128), Tb) * filt.to(torch.complex128)[:, None, :]
    H = H.reshape(nbands * B, K)
    (H.real * W.real.double() + H.imag * W.imag.double()).sum().backward()

    turns, _ = ops.zprep(z.to(DEV))
    Mdev = M.float().to(DEV)
    Q, QQ = ops.ortho_fwd(Mdev, True, True)
    coef, _ = ops.tf8_coefs(QQ, ig.to(DEV), b.to(DEV), c.to(DEV))
    cnow = cp.detach().float().to(DEV)
    Ts, _ = ops.tf8_tsave(turns, coef, delays.to(DEV), n, cnow, s.to(DEV), nbands, G, quad=False)
    assert rel_err(Ts.cpu().numpy(), T.detach().T.numpy()) < 5e-5
    grg = ops.tf_gain_grad(Ts, W.to(DEV), G, filt.to(DEV), nbands)
    assert rel_err(grg.cpu().numpy(), rg.grad.numpy()) < 1e-4
    part = ops.tf8_compose_bwd(turns, coef, delays.to(DEV), n, cnow, s.to(DEV), rgain.to(DEV), W.to(DEV), filt.to(DEV), nbands)
    gM, gb, gc = ops.tf8_param_grads(QQ, ig.to(DEV), part, bp.detach().float().to(DEV), cnow, Mdev, Q=Q)
    assert rel_err(gb.cpu().numpy(), bp.grad.numpy()) < 2e-4
    assert rel_err(gc.cpu().numpy(), cp.grad.numpy()) < 2e-4
    within(rel_err(gM.cpu().numpy(), Mr.grad.numpy()), 5e-5, "blocktf8 gM")
    # round 5: the forward pass leaves T' filt and 1 / Q beside T'; the adjoint pass that takes T' and 1 / Q back instead of
    # evaluating the two polynomials again accumulates the same records
    Ts2, _, Hg, Dinv = ops.tf8_tsave(turns, coef, delays.to(DEV), n, cnow, s.to(DEV), nbands, G, quad=False,
                                     filt=filt.to(DEV), want_H=True)
    assert torch.equal(Ts2, Ts)
    assert rel_err(Hg.cpu().numpy(), (Ts.view(nbands, G, K) * filt.to(DEV)[:, None, :]).reshape(nblk, K).cpu().numpy()) < 1e-6
    part2 = ops.tf8_compose_bwd(turns, coef, delays.to(DEV), n, cnow, s.to(DEV), rgain.to(DEV), W.to(DEV), filt.to(DEV),
                                nbands, saved=(Ts2, Dinv))
    assert rel_err(part2.sum(-1).cpu().numpy(), part.sum(-1).cpu().numpy()) < 1e-6
