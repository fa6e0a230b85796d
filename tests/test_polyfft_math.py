"""The identities csrc/polyfft.hip rests on, in numpy (no GPU), and the host-side planning around it.

A block's polynomial Q(z) = sum_S Q_S z^{m_S} with INTEGER degrees, evaluated on the reference's grid z_k = e^{2 pi i k / nfft},
k = 0 .. nfft / 2 (dataloader.py:552-566), is conj(rfft(q, nfft)) of the sparse real sequence q[m] = sum_{m_S = m mod nfft} Q_S;
the gradient of L = sum_k Re(conj(g_k) T_k), T = P / Q, with respect to the coefficients is a gather from two inverse real
transforms (weights nfft / 2, nfft at the two bins irfft counts once)."""
import numpy as np
import pytest
import torch


def _case(seed, nfft, nsub=64, dmax=None):
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, dmax or 3 * nfft, nsub)            # (degrees beyond nfft alias onto the grid's roots of unity)
    deg[5] = deg[9]                                          # two subsets of equal degree
    qs, ps = rng.standard_normal(nsub), rng.standard_normal(nsub)
    k = np.arange(nfft // 2 + 1)
    z = np.exp(2j * np.pi * k / nfft)
    return deg, qs, ps, z


@pytest.mark.parametrize("nfft", [64, 1024])
def test_polynomial_on_the_rfft_grid_is_a_real_transform(nfft):
    deg, qs, ps, z = _case(1, nfft)
    direct = (qs[None, :] * z[:, None] ** deg[None, :]).sum(1)
    seq = np.zeros(nfft)
    np.add.at(seq, deg % nfft, qs)
    assert np.abs(np.conj(np.fft.rfft(seq, nfft)) - direct).max() < 1e-9 * np.abs(direct).max()


@pytest.mark.parametrize("nfft", [64, 1024])
def test_coefficient_gradients_are_samples_of_inverse_transforms(nfft):
    deg, qs, ps, z = _case(2, nfft)
    rng = np.random.default_rng(3)
    K = nfft // 2 + 1
    g = rng.standard_normal(K) + 1j * rng.standard_normal(K)

    def loss(qc, pc):
        Q = (qc[None, :] * z[:, None] ** deg[None, :]).sum(1)
        P = (pc[None, :] * z[:, None] ** deg[None, :]).sum(1)
        return np.real(np.conj(g) * P / Q).sum()

    Q = (qs[None, :] * z[:, None] ** deg[None, :]).sum(1)
    P = (ps[None, :] * z[:, None] ** deg[None, :]).sum(1)
    T = P / Q
    u = np.conj(g) / Q
    v = -u * T
    w = np.full(K, nfft / 2.0)
    w[0] = w[-1] = nfft
    Gu = np.fft.irfft(w * u, nfft)
    Gv = np.fft.irfft(w * v, nfft)
    gp, gq = Gu[deg % nfft], Gv[deg % nfft]
    eps = 1e-6
    for S in (0, 5, 9, 17, 40):
        dq, dp = qs.copy(), ps.copy()
        dq[S] += eps
        dp[S] += eps
        num_q = (loss(dq, ps) - loss(qs, ps)) / eps
        num_p = (loss(qs, dp) - loss(qs, ps)) / eps
        assert abs(num_p - gp[S]) < 1e-4 * max(1.0, abs(gp[S])), (S, num_p, gp[S])
        assert abs(num_q - gq[S]) < 1e-4 * max(1.0, abs(gq[S])), (S, num_q, gq[S])


def test_plan_takes_integer_delays_only():
    from diffgfdn_amd import hip_ops as ops
    d = torch.tensor([641., 701., 809., 907., 1009., 1103., 1201., 1301.] * 3)
    T = ops.tfp_plan(d, 8, 131072)
    assert T is not None and T % 256 == 0 and 7672 < T <= 7672 + 256            # (largest degree 7672: that many + 1 samples, rounded up)
    assert ops.tfp_plan(d.clone(), 8, 4096) == 4096                               # (degrees alias: the whole grid)
    e = d.clone()
    e[3] += 0.25
    assert ops.tfp_plan(e, 8, 131072) is None
    assert ops.tfp_plan(d.clone(), 8, 131072 + 2) is None                         # (not a power of two)


def test_slot_table_is_a_permutation_of_the_bins():
    """tfp_slot_of_bin: column of bin k in the slot order of the odd-length transform (gfdn_irfft_odd_slot_order is host code:
    this needs the library, not a GPU)."""
    from diffgfdn_amd import hip_ops as ops
    n = 65537
    sob = ops.tfp_slot_of_bin(n, torch.device('cpu'))
    if sob is None:
        pytest.skip("no slot order for this length")
    bins, conj = ops.irfft_slot_order(n, torch.device('cpu'))
    Ku = (n + 1) // 2
    assert sob.dtype == torch.int32 and sob.numel() == Ku and int(sob[0]) == 0
    col = (sob.long() & 0x7fffffff)
    assert torch.equal(torch.sort(col).values, torch.arange(Ku))
    # column 1 + s holds bin bins[s], conjugated where conj[s]
    assert torch.equal(col[bins], torch.arange(1, Ku))
    assert torch.equal(sob[bins] < 0, conj)
