"""Blocks of up to nine delay lines in polynomial form (csrc/blocktf9.hip + csrc/polyfft.hip): coefficient records, the
group responses on the reference's grid by transforms and the adjoint, against the per-bin elimination path
(functional.ResolventSolve + GroupSums: reference feedback_loop.py:326-391, model.py:209-252) and float64 torch."""
import numpy as np
import pytest
from tests.margins import within
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(n, nblk, nfft, seed, dmax):
    g = torch.Generator().manual_seed(seed)
    M = ((2 * torch.rand(nblk, n, n, generator=g) - 1) / np.sqrt(n)).to(DEV)
    b = (0.3 + torch.rand(nblk * n, generator=g)).to(DEV)
    c = (0.3 + torch.rand(nblk * n, generator=g)).to(DEV)
    delays = torch.randint(7, dmax, (nblk * n,), generator=g).to(torch.float32).to(DEV)
    K = nfft // 2 + 1
    z = torch.tensor(np.exp(2j * np.pi * np.arange(K) / nfft), device=DEV)
    return M, b, c, delays, z, K


def _S_ref64(M, b, c, delays, z):
    """S (nblk, K) complex128 by per-bin solves"""
    nblk, n, _ = M.shape
    out = []
    for q in range(nblk):
        sl = slice(q * n, (q + 1) * n)
        D = torch.diag_embed(z[:, None] ** delays[sl].double()[None, :])
        X = D - M[q].double().to(torch.complex128)[None]
        y = torch.linalg.solve(X, b[sl].double().to(torch.complex128)[None, :, None].expand(z.numel(), n, 1))
        out.append((c[sl].double().to(torch.complex128)[None, :] * y[..., 0]).sum(-1))
    return torch.stack(out)


@pytest.mark.parametrize("n,nfft", [(9, 2048), (7, 1024), (9, 131072)])
def test_group_responses_by_transforms_and_their_adjoint(n, nfft):
    from diffgfdn_amd import hip_ops as ops
    from diffgfdn_amd.functional import SubFdnTransforms
    nblk = 3
    M, b, c, delays, z, K = _case(n, nblk, nfft, 3 + n, 2400 if nfft > 4096 else nfft // 7)
    T_seq = ops.tfp_plan(delays, n, nfft)
    assert T_seq is not None
    Mp, bp, cp = (t.clone().requires_grad_(True) for t in (M, b, c))
    S = SubFdnTransforms.apply(Mp, bp, cp, delays, n, nfft, T_seq)
    sel = torch.arange(0, K, 1 if nfft <= 4096 else 97, device=DEV)
    ref = _S_ref64(M, b, c, delays, z[sel])
    assert ((S[:, sel].to(torch.complex128) - ref).abs().max() / ref.abs().max()) < 5e-5
    # adjoint against float64 autograd of the per-bin solve (on the sampled bins: a loss that only reads those)
    g = torch.Generator().manual_seed(1)
    wgt = torch.complex(torch.randn(nblk, sel.numel(), generator=g), torch.randn(nblk, sel.numel(), generator=g)).to(DEV)
    L = (S[:, sel] * wgt.to(torch.complex64)).real.sum()
    L.backward()
    M64, b64, c64 = (t.double().clone().requires_grad_(True) for t in (M, b, c))
    S64 = []
    for q in range(nblk):
        slq = slice(q * n, (q + 1) * n)
        D = torch.diag_embed(z[sel][:, None] ** delays[slq].double()[None, :])
        X = D - M64[q].to(torch.complex128)[None]
        y = torch.linalg.solve(X, b64[slq].to(torch.complex128)[None, :, None].expand(sel.numel(), n, 1))
        S64.append((c64[slq].to(torch.complex128)[None, :] * y[..., 0]).sum(-1))
    L64 = (torch.stack(S64) * wgt.to(torch.complex128)).real.sum()
    L64.backward()
    for got, want, tol in ((Mp.grad, M64.grad, 3e-4), (bp.grad, b64.grad, 3e-4), (cp.grad, c64.grad, 3e-4)):
        assert (got.double() - want).abs().max() < tol * want.abs().max(), ((got.double() - want).abs().max(), want.abs().max())


def test_directional_model_branch_equals_elimination_path():
    """DiffDirectionalFDNVarReceiverPos.sub_fdn_group_sums on the transform path == the per-bin elimination path, values and
    gradients (3 groups x 9 lines, nfft 8192)."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffDirectionalFDNVarReceiverPos
    fs, nfft, G, L = 8000.0, 8192, 3, 9
    delays = [173, 181, 191, 193, 197, 199, 211, 223, 227, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283, 293, 307,
              311, 313, 317, 331]
    torch.manual_seed(0)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    net = DiffDirectionalFDNVarReceiverPos(fs, G, delays, torch.device(DEV), fl, of, ambi_order=2,
                                           common_decay_times=np.linspace(0.2, 0.5, G)[None, :], use_colorless_loss=True,
                                           analysis_matrix=np.random.RandomState(0).randn(12, 9)).to(DEV)
    net.per_delay_output = False
    z = torch.tensor(np.exp(1j * 2 * np.pi * np.fft.rfftfreq(nfft)), device=DEV)
    wgt = torch.randn(G, nfft // 2 + 1, device=DEV)
    res = {}
    for fft in (True, False):
        net.sub_fdn_by_transforms = fft
        net.zero_grad(set_to_none=True)
        S, Y = net.sub_fdn_group_sums(z)
        assert (Y is None) == fft
        ((S.abs() ** 2) * wgt).sum().backward()
        res[fft] = (S.detach().clone(), net.feedback_loop.M.grad.clone(), net.input_gains.grad.clone(),
                    net.output_gains.grad.clone())
    assert (res[True][0] - res[False][0]).abs().max() < 5e-5 * res[False][0].abs().max()
    for a, b_ in zip(res[True][1:], res[False][1:]):
        within(float((a - b_).abs().max() / b_.abs().max()), 5e-5, "blocktf9")
