"""CPU oracle for the DiffGFDN hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file restates, op for op and cast for cast, what the reference
(orchidas/DiffGFDN, /root/reference) computes on the frequency-sampled GFDN path:
feedback-matrix parameterisation, per-bin resolvent, transfer functions of the three
model variants, the sub-FDN (colorless) responses, the EDR / EDC / directional-EDC /
colorless losses and one optimiser step of the grid trainer.  Every function cites the
reference file:line it follows.  Gradients come from torch autograd on this forward.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it -- as the checker / CPU baseline, never as the thing shipped or measured on the
GPU.  The product package ``diffgfdn_amd`` must not import anything from ``oracle/``.

Parity pinning: the reference's own tests hold no vectors for this path,
so the oracle is pinned against outputs of the reference itself, imported in the build
container by ``tests/golden/gen_golden.py`` and committed as ``tests/golden/*.npz``
(``tests/test_oracle_golden.py`` checks every one).  Pieces whose reference source is an
absent third-party package (slope2noise.decay_kernel envelopes, spaudiopy analysis
matrix, pyfar FIR taps) are taken as INPUT arrays: parity is unpinned for those inputs
themselves and pinned for everything downstream of them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

F32_EPS = float(torch.finfo(torch.float32).eps)


# --------------------------------------------------------------------------------------
# small helpers  (reference: src/diff_gfdn/utils.py)
# --------------------------------------------------------------------------------------
def db(x: torch.Tensor, is_squared: bool = False, min_value: float = -200.0) -> torch.Tensor:
    """utils.py:16-40 -- factor*log10(|x| + eps_f32), clipped below at min_value."""
    factor = 10.0 if is_squared else 20.0
    y = factor * torch.log10(torch.abs(x) + F32_EPS)
    return y.clip(min=min_value)


def db2lin(x, is_squared: bool = False):
    """utils.py:43-59."""
    e = 0.1 if is_squared else 0.05
    if torch.is_tensor(x):
        return torch.pow(10.0, x * e)
    return np.power(10.0, np.asarray(x) * e)


def ms_to_samps(ms: float, fs: float) -> int:
    """utils.py:62-80 (scalar branch: int() truncation)."""
    return int(ms * 1e-3 * fs)


def to_complex(x: torch.Tensor) -> torch.Tensor:
    """utils.py:144-146."""
    return torch.complex(x, torch.zeros_like(x))


def decay_times_to_gain_per_sample(t60, delays, fs):
    """absorption_filters.py:40-53 -- gamma = 10^(0.05 * (-60 m / (fs T60)))."""
    if torch.is_tensor(t60):
        return db2lin(-60 * delays / (fs * t60))
    return db2lin(-60 * np.asarray(delays) / (fs * t60))


# --------------------------------------------------------------------------------------
# feedback-matrix parameterisation  (reference: src/diff_gfdn/feedback_loop.py)
# --------------------------------------------------------------------------------------
def skew(X: torch.Tensor) -> torch.Tensor:
    """feedback_loop.py:16-25."""
    A = X.triu(1)
    return A - A.transpose(-1, -2)


def ortho_param(X: torch.Tensor) -> torch.Tensor:
    """feedback_loop.py:270 -- expm(skew(X))."""
    return torch.matrix_exp(skew(X))


def nd_unitary(alpha: torch.Tensor, N: int):
    """feedback_loop.py:39-87 -- recursive Givens product; N == 1 returns python int 1."""
    assert len(alpha) == N * (N - 1) // 2
    if N == 1:
        return 1
    rot = torch.eye(N, dtype=alpha.dtype)
    start = (N - 1) * (N - 2) // 2
    cur = alpha[start:]
    for i in range(N - 1):
        R = torch.eye(N, dtype=alpha.dtype)
        R[i, i] = torch.cos(cur[i])
        R[i, -1] = -torch.sin(cur[i])
        R[-1, i] = torch.sin(cur[i])
        R[-1, -1] = torch.cos(cur[i])
        rot = torch.mm(R, rot)
    big = torch.eye(N, dtype=alpha.dtype)
    big[:N - 1, :N - 1] = nd_unitary(alpha[:start], N - 1)
    return torch.mm(rot, big)


def block_mixing_matrix(M: torch.Tensor) -> torch.Tensor:
    """feedback_loop.py:393-404 -- block (i,j) = Q_i @ Q_j (diagonal blocks are Q_i^2)."""
    G, n, _ = M.shape
    Q = [ortho_param(M[i]) for i in range(G)]
    out = torch.zeros((G * n, G * n), dtype=M.dtype)
    for i in range(G):
        for j in range(G):
            out[i * n:(i + 1) * n, j * n:(j + 1) * n] = torch.mm(Q[i], Q[j])
    return out


def coupled_feedback_matrix(M: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """feedback_loop.py:406-411, 424-455 (SCALAR coupling) -- A = block_M o kron(Phi, 1)."""
    G, n, _ = M.shape
    block_M = block_mixing_matrix(M)
    phi = nd_unitary(alpha.clamp(min=-np.pi, max=np.pi), G)
    if G == 1:
        # torch.kron(1, ones) raises in the reference -> falls back to block_M (:441-445)
        return block_M
    return block_M * torch.kron(phi, torch.ones((n, n), dtype=M.dtype))


def feedback_loop_forward(z: torch.Tensor, delays: torch.Tensor, gamma: torch.Tensor,
                          A: torch.Tensor) -> torch.Tensor:
    """feedback_loop.py:326-391, scalar gains & scalar/random coupling.

    z (K,) c128, delays (N,) f32, gamma (N,) real, A (N,N) real f32.
    Returns P (K,N,N) complex64 = inv(diag(z^m)/gamma - A) computed in c128."""
    K = len(z)
    D = torch.diag_embed(torch.unsqueeze(z, dim=-1) ** delays)            # :330
    Gamma = to_complex(torch.diag(gamma))                                  # :347
    Acplx = to_complex(A).unsqueeze(0).repeat(K, 1, 1)                     # :359
    Gamma_inv = torch.diag(1.0 / torch.diagonal(Gamma))                    # :385
    Ddecay = D * Gamma_inv.unsqueeze(0).repeat(K, 1, 1)                    # :386
    return torch.linalg.inv(Ddecay - Acplx).to(torch.complex64)            # :391


def matrix_convolution(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """utils.py:216-239 -- polynomial-matrix product (M, N, K) x (N, Q, R) -> (M, Q, K + R - 1); the reference
    convolves entry by entry with torchaudio.functional.convolve(mode='full') (torchaudio is absent here: the
    full linear convolution is written out)."""
    M, N, K = A.shape
    _, Q, R = B.shape
    C = torch.zeros((M, Q, K + R - 1), dtype=A.dtype)
    for row in range(M):
        for col in range(Q):
            for it in range(N):
                for i in range(K):
                    C[row, col, i:i + R] = C[row, col, i:i + R] + A[row, it, i] * B[it, col, :]
    return C


def fir_paraunitary(unitary_matrix: torch.Tensor, unit_vectors: torch.Tensor) -> torch.Tensor:
    """feedback_loop.py:90-143 -- cascade of order-1 Householder factors I - (1 - z^-1) v v^T (one per column of
    ``unit_vectors`` (N, order - 1)), then the unitary zeroth-order factor -> (N, N, order)."""
    N = unitary_matrix.shape[0]
    poly = torch.eye(N, dtype=unit_vectors.dtype)[..., None]
    for k in range(unit_vectors.shape[1]):
        vv = torch.outer(unit_vectors[:, k], unit_vectors[:, k])
        house = torch.stack([torch.eye(N, dtype=vv.dtype) - vv, vv], dim=-1)        # :106-117
        poly = matrix_convolution(house, poly)
    return matrix_convolution(poly, unitary_matrix.reshape(N, N, 1))


def filter_coupling_matrix(raw_unitary: torch.Tensor, raw_unit_vectors: torch.Tensor) -> torch.Tensor:
    """feedback_loop.py:413-420 -- FILTER coupling Phi (G, G, order) from the raw parameters."""
    v = raw_unit_vectors / (torch.norm(raw_unit_vectors, dim=0, keepdim=True) + 1e-9)
    return fir_paraunitary(ortho_param(raw_unitary), v)


def filter_coupled_feedback_matrix(M: torch.Tensor, phi: torch.Tensor) -> torch.Tensor:
    """feedback_loop.py:447-455 -- A[..., p] = block_M o kron(Phi_p, 1), (N, N, order) complex."""
    G, n, _ = M.shape
    block_M = block_mixing_matrix(M)
    ones = torch.ones((n, n), dtype=M.dtype)
    A = torch.stack([block_M * torch.kron(phi[..., p].contiguous(), ones) for p in range(phi.shape[-1])], dim=-1)
    return to_complex(A)


def feedback_loop_forward_filter(z: torch.Tensor, delays: torch.Tensor, gamma: torch.Tensor,
                                 A_poly: torch.Tensor) -> torch.Tensor:
    """feedback_loop.py:326-391 with FILTER coupling (:362-373): A(z_k) = sum_p A_p z_k^-p.
    A_poly (N, N, order) complex64 -> P (K, N, N) complex64."""
    K = len(z)
    order = A_poly.shape[-1]
    D = torch.diag_embed(torch.unsqueeze(z, dim=-1) ** delays)
    zp = (z.view(-1, 1) ** -torch.arange(0, order)).permute(1, 0)                  # (order, K)
    A = torch.einsum('jim,mn->jimn', A_poly, zp)
    A = torch.sum(A, dim=2).permute(2, 0, 1).to(torch.complex64)
    Gamma_inv = torch.diag(1.0 / torch.diagonal(to_complex(torch.diag(gamma))))
    Ddecay = D * Gamma_inv.unsqueeze(0).repeat(K, 1, 1)
    return torch.linalg.inv(Ddecay - A).to(torch.complex64)


def svf_cutoff_freqs(fs: float) -> torch.Tensor:
    """filters/geq.py:9-56 (eq_freqs defaults) + gain_filters.py:299-303 / model.py:711-716."""
    centre, c = [], 31.25
    while c < 16000:
        centre.append(c * np.power(2, 1.0))
        c = centre[-1]
    centre = torch.tensor(centre)
    sc = torch.tensor([centre[0] / np.power(2, 0.5), centre[-1] * np.power(2, 0.5)])
    return torch.pi * torch.cat((torch.tensor([sc[0]]), centre, torch.tensor([sc[-1]]))) / fs


def svf_biquad_cascade(cutoffs: torch.Tensor, params: torch.Tensor, cpf: float):
    """One cascade: params (S, 2) = constrained [resonance, gain dB] -> (num (S,3), den (S,3)) float32
    (gain_filters.py:36-103 SVF mixing coefficients, :117-151 from_svf_coeffs)."""
    S = params.shape[0]
    num = torch.zeros((S, 3))
    den = torch.zeros((S, 3))
    for i in range(S):
        f, R = cutoffs[i], params[i, 0]
        G = db2lin(params[i, 1])
        if i == 0:
            m_lp, m_bp, m_hp = G, 2 * R * torch.sqrt(G), torch.ones_like(G)          # lowshelf
        elif i == S - 1:
            m_lp, m_bp, m_hp = torch.ones_like(G), 2 * R * torch.sqrt(G), G          # highshelf
        else:
            m_lp, m_bp, m_hp = torch.ones_like(G), 2 * R * G, torch.ones_like(G)     # peaking
        num[i, 0] = f ** 2 * m_lp + f * m_bp + m_hp
        num[i, 1] = (2 * f ** 2 * m_lp - 2 * m_hp) * cpf
        num[i, 2] = (f ** 2 * m_lp - f * m_bp + m_hp) * cpf ** 2
        den[i, 0] = f ** 2 + 2 * R * f + 1
        den[i, 1] = (2 * f ** 2 - 2) * cpf
        den[i, 2] = (f ** 2 - 2 * R * f + 1) * cpf ** 2
    return num, den


def svf_group_responses(z: torch.Tensor, fs: float, raw_params: torch.Tensor, cpf: float = 1.0) -> torch.Tensor:
    """raw (..., S, 2) unconstrained -> (..., K) complex64 cascade responses: scaled sigmoids (gain_filters.py:
    327-330 / model.py:733-737, :853-866), SVF -> biquads, SOSFilter.forward (:221-241)."""
    cut = svf_cutoff_freqs(fs)
    lead = raw_params.shape[:-2]
    flat = raw_params.reshape(-1, raw_params.shape[-2], 2)
    out = []
    for q in range(flat.shape[0]):
        prm = torch.stack([scaled_sigmoid(flat[q, :, 0], 1e-6, 1.0), scaled_sigmoid(flat[q, :, 1], -6.0, 6.0)], dim=-1)
        num, den = svf_biquad_cascade(cut, prm, cpf)
        out.append(sos_response(z, torch.stack((num, den), dim=-1).unsqueeze(0))[0])
    return torch.stack(out).reshape(*lead, len(z))


def sos_response(z: torch.Tensor, coeffs: torch.Tensor) -> torch.Tensor:
    """gain_filters.py:221-241 (SOSFilter.forward) for N cascades: coeffs (N, S, 3, 2), [..., 0] numerator,
    [..., 1] denominator -> (N, K) complex64 (the reference accumulates the product in complex64)."""
    N, S = coeffs.shape[:2]
    out = []
    for n in range(N):
        H = torch.ones(len(z), dtype=torch.complex64)
        for k in range(S):
            b, a = coeffs[n, k, :, 0], coeffs[n, k, :, 1]
            H *= torch.div(b[0] + b[1] * torch.pow(z, -1) + b[2] * torch.pow(z, -2),
                           a[0] + a[1] * torch.pow(z, -1) + a[2] * torch.pow(z, -2))
        out.append(H)
    return torch.stack(out)


def feedback_loop_forward_absorption(z: torch.Tensor, delays: torch.Tensor, coeffs: torch.Tensor,
                                     A: torch.Tensor) -> torch.Tensor:
    """feedback_loop.py:326-391 with use_absorption_filters (SOS list branch :335-341, :376-381):
    P (K,N,N) c64 = inv(diag(z^m) Gamma(z)^-1 - A)."""
    K = len(z)
    N = len(delays)
    D = torch.diag_embed(torch.unsqueeze(z, dim=-1) ** delays)
    Gamma = torch.zeros((N, N, K), dtype=torch.complex64)
    G = sos_response(z, coeffs)
    for k in range(N):
        Gamma[k, k, :] = G[k]
    Acplx = to_complex(A).unsqueeze(0).repeat(K, 1, 1)
    Gamma_inv = torch.diag_embed(1.0 / torch.diagonal(Gamma), dim1=0, dim2=1)
    Ddecay = D * Gamma_inv.permute(-1, 0, 1)
    return torch.linalg.inv(Ddecay - Acplx).to(torch.complex64)


def feedback_loop_forward_filter_absorption(z: torch.Tensor, delays: torch.Tensor, coeffs: torch.Tensor,
                                            A_poly: torch.Tensor) -> torch.Tensor:
    """feedback_loop.py:326-391 with BOTH frequency-dependent parts: absorption filters on the lines (SOS list
    branch :335-341, Gamma_inv per bin :376-381) and FILTER coupling A(z_k) = sum_p A_p z_k^-p (:362-373).
    coeffs (N, S, 3, 2), A_poly (N, N, order) complex64 -> P (K, N, N) complex64."""
    K, N = len(z), len(delays)
    order = A_poly.shape[-1]
    D = torch.diag_embed(torch.unsqueeze(z, dim=-1) ** delays)
    Gamma = torch.zeros((N, N, K), dtype=torch.complex64)
    G = sos_response(z, coeffs)
    for k in range(N):
        Gamma[k, k, :] = G[k]
    zp = (z.view(-1, 1) ** -torch.arange(0, order)).permute(1, 0)
    A = torch.einsum('jim,mn->jimn', A_poly, zp)
    A = torch.sum(A, dim=2).permute(2, 0, 1).to(torch.complex64)
    Gamma_inv = torch.diag_embed(1.0 / torch.diagonal(Gamma), dim1=0, dim2=1)
    Ddecay = D * Gamma_inv.permute(-1, 0, 1)
    return torch.linalg.inv(Ddecay - A).to(torch.complex64)


# --------------------------------------------------------------------------------------
# models  (reference: src/diff_gfdn/model.py)
# --------------------------------------------------------------------------------------
def sub_fdn_output(z: torch.Tensor, M: torch.Tensor, input_gains: torch.Tensor,
                   output_gains: torch.Tensor, delays: torch.Tensor
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """model.py:209-252 -- raw M_g (not expm), no absorption.  Returns Hout (K,G) c64,
    Hout_per_del (N,K,G) c64."""
    G, n, _ = M.shape
    K = len(z)
    N = G * n
    Hout = torch.zeros((K, G), dtype=torch.complex64)
    Hpd = torch.zeros((N, K, G), dtype=torch.complex64)
    for k in range(G):
        idx = torch.arange(k * n, (k + 1) * n)
        C = to_complex(output_gains[idx].expand(n, K))
        B = to_complex(input_gains[idx].expand(n, K))
        Ak = M[k].unsqueeze(0).repeat(K, 1, 1)
        D = torch.diag_embed(torch.unsqueeze(z, dim=-1) ** delays[idx])
        P = torch.linalg.inv(D - Ak).to(torch.complex64)
        H_tmp = torch.einsum('kn, knm -> knm', C.permute(1, 0), P).permute(1, -1, 0)
        Hpd[idx, :, k] = torch.einsum('nmk, mk -> nk', H_tmp, B)
        H = torch.einsum('kn, knm -> km', C.permute(1, 0), P).permute(1, 0)
        Hout[..., k] = torch.einsum('mk, mk -> k', H, B)
    return Hout, Hpd


def var_receiver_forward(z: torch.Tensor, input_gains: torch.Tensor,
                         output_gains: torch.Tensor, receiver_gains: torch.Tensor,
                         P: torch.Tensor, direct: torch.Tensor, n_per_group: int
                         ) -> torch.Tensor:
    """model.py:569-619 (DiffGFDNVarReceiverPos.forward, MLP-gain branch).

    receiver_gains (B,G) = sigmoid-scaled MLP output; P (K,N,N) c64; direct (B,K) c128."""
    Bsz = receiver_gains.shape[0]
    N = input_gains.shape[0]
    K = len(z)
    C_init = to_complex(output_gains.expand(Bsz, N, K))                     # :583-585
    expanded = receiver_gains.repeat_interleave(n_per_group, dim=1)         # gain_filters.py:526
    Cg = expanded.unsqueeze(-1).repeat(1, 1, K)                             # gain_filters.py:530
    C = to_complex(Cg) * C_init                                             # :592
    Bm = to_complex(input_gains.expand(Bsz, N, K))                          # :608-610
    Htemp = torch.einsum('knb, knm -> kmb', C.permute(-1, 1, 0), P).permute(-1, 1, 0)
    return torch.einsum('bmk, bmk -> bk', Htemp, Bm) + direct              # :619


def var_source_receiver_forward(z: torch.Tensor, input_gains: torch.Tensor, output_gains: torch.Tensor,
                                receiver_gains: torch.Tensor, source_gains: torch.Tensor,
                                P: torch.Tensor, direct: torch.Tensor, n_per_group: int) -> torch.Tensor:
    """model.py:402-452 (DiffGFDNVarSourceReceiverPos.forward, MLP-gain branches on both sides).

    receiver_gains, source_gains (B,G) = sigmoid-scaled outputs of the two MLPs (gain_filters.py:497-534; the
    input-side network reads x['source_position'], :503-505); P (K,N,N) c64; direct (B,K) c128."""
    Bsz = receiver_gains.shape[0]
    N = input_gains.shape[0]
    K = len(z)
    C_init = to_complex(output_gains.expand(Bsz, N, K))                     # :416-418
    B_init = to_complex(input_gains.expand(Bsz, N, K))                      # :419-421
    rg = receiver_gains.repeat_interleave(n_per_group, dim=1).unsqueeze(-1).repeat(1, 1, K)
    sg = source_gains.repeat_interleave(n_per_group, dim=1).unsqueeze(-1).repeat(1, 1, K)
    C = to_complex(rg) * C_init                                             # :427
    Bm = to_complex(sg) * B_init                                            # :432
    Htemp = torch.einsum('knb, knm -> kmb', C.permute(-1, 1, 0), P).permute(-1, 1, 0)   # :437-438
    return torch.einsum('bmk, bmk -> bk', Htemp, Bm) + direct              # :444


def var_source_receiver_forward_filters(z: torch.Tensor, input_gains: torch.Tensor, output_gains: torch.Tensor,
                                        receiver_side: torch.Tensor, source_side: torch.Tensor, P: torch.Tensor,
                                        direct: torch.Tensor, n_per_group: int) -> torch.Tensor:
    """model.py:402-452 (DiffGFDNVarSourceReceiverPos.forward) with SVF filters on either side (constructor branches
    :347-400): ``receiver_side`` / ``source_side`` are the per-group factors of C / B -- (B, G, K) complex cascade responses
    of SVF_from_MLP (gain_filters.py:334-402; every delay line of a group carries its group's filter, :396-400) or (B, G)
    real MLP gains (:497-534).  P (K,N,N) c64; direct (B,K) c128."""
    Bsz, N, K = receiver_side.shape[0], input_gains.shape[0], len(z)
    C_init = to_complex(output_gains.expand(Bsz, N, K))                     # :416-418
    B_init = to_complex(input_gains.expand(Bsz, N, K))                      # :419-421

    def per_line(f):
        if f.dim() == 2:                                                    # scalar gains: :427 / :432
            return to_complex(f.repeat_interleave(n_per_group, dim=1).unsqueeze(-1).repeat(1, 1, K))
        return f.repeat_interleave(n_per_group, dim=1)                      # filters: :425 / :430
    C = per_line(receiver_side) * C_init
    Bm = per_line(source_side) * B_init
    Htemp = torch.einsum('knb, knm -> kmb', C.permute(-1, 1, 0), P).permute(-1, 1, 0)   # :437-438
    return torch.einsum('bmk, bmk -> bk', Htemp, Bm) + direct              # :444


def colorless_fdn_forward(z: torch.Tensor, delays: torch.Tensor, gamma: torch.Tensor,
                          input_gains: torch.Tensor, output_gains: torch.Tensor, W: torch.Tensor):
    """colorless_fdn/model.py:63-92 (ColorlessFDN.forward): one group, dense feedback matrix
    Q = ortho_param(W) (CouplingMatrixType.RANDOM, feedback_loop.py:350-352).  -> (H (K,), H_per_del (N, K))."""
    N = input_gains.shape[0]
    K = len(z)
    P = feedback_loop_forward(z, delays, gamma, ortho_param(W))
    C = to_complex(output_gains.expand(N, K))                               # :74-75
    Bm = to_complex(input_gains)                                            # :78
    Htemp = torch.einsum('kn, knm -> km', C.permute(-1, 0), P)              # :84
    H = torch.einsum('ik, kj -> ij', Htemp, Bm).squeeze()                   # :86
    H_tmp = torch.einsum('kn, knm -> knm', C.permute(1, 0), P).permute(1, -1, 0)   # :89-90
    H_per_del = torch.einsum('nmk, mk -> nk', H_tmp, Bm)                    # :91
    return H, H_per_del


def single_pos_forward(z, input_gains, output_gains, input_scalars, output_scalars, P,
                       direct, n_per_group):
    """model.py:779-836 (DiffGFDNSinglePos.forward, scalar in/out branch)."""
    N = input_gains.shape[0]
    K = len(z)
    G = output_scalars.shape[0]
    C_init = to_complex(output_gains.expand(N, K))
    B_init = to_complex(input_gains.expand(N, K))
    C = to_complex(output_scalars.expand(G, K).repeat_interleave(n_per_group, dim=0)) * C_init
    Bm = to_complex(input_scalars.expand(G, K).repeat_interleave(n_per_group, dim=0)) * B_init
    Htemp = torch.einsum('kn, knm -> km', C.permute(-1, 0), P)
    H = torch.einsum('ki, ik -> k', Htemp, Bm)
    return H + direct


def directional_forward(z, input_gains, output_gains, sh_gains, P, G: int, n_per_group: int):
    """model.py:1043-1088 (DiffDirectionalFDNVarReceiverPos.forward).

    sh_gains (B,G,n_per_group) already L2-normalised (spatial_sampling/model.py:78-80).
    Note Htemp = einsum('knm,bnk->bmk') contracts P's FIRST index (P^T b); no d(z)."""
    Bsz = sh_gains.shape[0]
    N = input_gains.shape[0]
    K = len(z)
    cur = output_gains.reshape(G, n_per_group).unsqueeze(0).unsqueeze(-1)
    C_init = to_complex(cur.expand(Bsz, G, n_per_group, K))
    sg = sh_gains.reshape(Bsz, G, n_per_group, 1).repeat(1, 1, 1, K)
    C = to_complex(sg) * C_init
    Bm = to_complex(input_gains.expand(Bsz, N, K))
    Htemp = torch.einsum('knm, bnk -> bmk', P, Bm).reshape(Bsz, G, n_per_group, K)
    return (C * Htemp).sum(dim=1)


def normalise_sh_weights(w: torch.Tensor) -> torch.Tensor:
    """spatial_sampling/model.py:78-80."""
    return w / (torch.norm(w, dim=-1, keepdim=True) + 1e-6)


def sh_to_directional(analysis_matrix: torch.Tensor, H_sh: torch.Tensor) -> torch.Tensor:
    """trainer.py:853-865 -- einsum('jl,blk->bjk')."""
    return torch.einsum('jl, blk -> bjk', to_complex(analysis_matrix), H_sh)


# --------------------------------------------------------------------------------------
# position -> gain MLP  (reference: src/diff_gfdn/dnn.py, gain_filters.py)
# --------------------------------------------------------------------------------------
def sinusoidal_encoding(pos: torch.Tensor, num_fourier_features: int) -> torch.Tensor:
    """dnn.py:89-126 -- output tensor is float32 (torch.zeros default), freqs log-spaced 1..32."""
    P, F = pos.shape
    enc = torch.zeros(P, F * num_fourier_features * 2)
    freqs = torch.exp(torch.linspace(np.log(1.0), np.log(32.0), num_fourier_features))
    s = 0
    for k in range(num_fourier_features):
        enc[:, s:s + 2 * F] = torch.cat((torch.sin(freqs[k] * np.pi * pos),
                                         torch.cos(freqs[k] * np.pi * pos)), dim=-1)
        s += 2 * F
    return enc


def mlp_forward(x: torch.Tensor, weights: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                norms: Sequence[Tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """dnn.py:331-400 -- [Linear, LayerNorm, ReLU] x (1+hidden), then Linear."""
    h = x
    for i, (W, b) in enumerate(weights[:-1]):
        h = torch.nn.functional.linear(h, W, b)
        g, beta = norms[i]
        h = torch.nn.functional.layer_norm(h, (h.shape[-1],), g, beta)
        h = torch.relu(h)
    W, b = weights[-1]
    return torch.nn.functional.linear(h, W, b)


def scaled_sigmoid(x, lo: float, hi: float):
    """dnn.py:21-36."""
    return lo + (hi - lo) * (1.0 / (1 + torch.exp(-x)))


# --------------------------------------------------------------------------------------
# losses  (reference: src/diff_gfdn/losses.py, colorless_fdn/losses.py)
# --------------------------------------------------------------------------------------
def stft_onesided(rir: torch.Tensor, win_size: int, hop_size: int) -> torch.Tensor:
    """losses.py:501-535 -- zero-pad to a multiple of hop, periodic Hann (float32 window,
    as torch.hann_window default), center=False, one-sided, un-normalised."""
    T = rir.shape[-1]
    if T % hop_size != 0:
        extra = hop_size * int(np.ceil(T / hop_size)) - T
        rir = torch.nn.functional.pad(rir, (0, extra))
    window = torch.hann_window(win_size)
    return torch.stft(rir, win_size, hop_length=hop_size, win_length=win_size,
                      window=window, center=False, normalized=False, onesided=True,
                      return_complex=True)


def edr_from_stft(S: torch.Tensor) -> torch.Tensor:
    """losses.py:556-575 -- float32 buffer, frame loop of tail sums, db(is_squared)."""
    edr = torch.zeros(S.shape, dtype=torch.float32)
    for m in range(S.shape[-1]):
        edr[..., m] = torch.sum(torch.abs(S[..., m:]) ** 2, axis=-1)
    return db(edr, is_squared=True)


def edr_loss(target: torch.Tensor, achieved: torch.Tensor, win_size: int = 4096,
             hop_size: int = 2048, reduced_pole_radius: Optional[float] = None,
             freq_weights: Optional[torch.Tensor] = None,
             erb_filters: Optional[torch.Tensor] = None) -> torch.Tensor:
    """losses.py:430-495 -- irfft(n = K) (!), STFT, EDR in dB, per-item normalised L1, summed.
    ``erb_filters`` (bands, F): the ERB grouping of :545-551 -- the STFT is replaced by erb_filters |S| (einsum
    'nk,bkt->bnt' on the magnitudes) before the EDR."""
    t_rir = torch.fft.irfft(target, target.shape[-1])
    a_rir = torch.fft.irfft(achieved, achieved.shape[-1])
    if reduced_pole_radius is not None:
        a_rir = a_rir * torch.pow(1.0 / reduced_pole_radius, torch.arange(0, a_rir.shape[-1]))
    St, Sa = stft_onesided(t_rir, win_size, hop_size), stft_onesided(a_rir, win_size, hop_size)
    if erb_filters is not None:
        eq = 'nk,kt->nt' if St.ndim == 2 else 'nk,bkt->bnt'
        St = torch.einsum(eq, erb_filters.to(torch.abs(St).dtype), torch.abs(St))
        Sa = torch.einsum(eq, erb_filters.to(torch.abs(Sa).dtype), torch.abs(Sa))
    t_edr = edr_from_stft(St)
    a_edr = edr_from_stft(Sa)
    freq_loss = torch.sum(torch.abs(t_edr - a_edr), dim=-1)
    if freq_weights is not None:
        freq_loss = freq_loss * freq_weights
    if t_edr.ndim == 3:
        per_item = torch.div(torch.sum(freq_loss, dim=-1),
                             torch.sum(torch.abs(t_edr), dim=[-1, -2]))
        return torch.sum(per_item)
    return torch.div(torch.sum(freq_loss), torch.sum(torch.abs(t_edr)))


def edr_frequency_weights(sample_rate: float, win_size: int) -> torch.Tensor:
    """losses.py:419-428, 49-57 -- inverse scaled sigmoid, 2 below ~1 kHz -> 1 above.

    NB the reference passes (bottom, top) swapped into (top, bottom) slots: reproduced."""
    freqs = torch.tensor(np.fft.rfftfreq(win_size, d=1.0 / sample_rate))
    scale, cutoff, top, bottom = 10 ** (-2.5), 1e3, 2.0, 1.0
    # scaled_shifted_sigmoid_inverse(x, scale_factor, cutoff, top=bottom_arg, bottom=top_arg)
    t_, b_ = bottom, top
    return b_ + torch.div((t_ - b_), (1 + torch.exp(scale * (freqs - cutoff))))


def schroeder(signal: torch.Tensor) -> torch.Tensor:
    """losses.py:187-199."""
    return torch.flip(torch.cumsum(torch.flip(signal ** 2, dims=[-1]), dim=-1), dims=[-1])


def edc_loss(target: torch.Tensor, achieved: torch.Tensor, max_ir_len_samps: int,
             mixing_time_samps: int, mask_index: Optional[torch.Tensor] = None
             ) -> torch.Tensor:
    """losses.py:201-238 (broadband branch).  mask_index: kept time indices (the reference
    draws argwhere(bernoulli(U(0,1))) of shape (M,1) -- pass the same tensor to reproduce)."""
    L = min(max_ir_len_samps, target.shape[-1])
    t_rir = torch.fft.irfft(target, target.shape[-1])[..., mixing_time_samps:L]
    a_rir = torch.fft.irfft(achieved, achieved.shape[-1])[..., mixing_time_samps:L]
    t_edc = schroeder(t_rir)
    a_edc = schroeder(a_rir)
    if mask_index is None:
        mask_index = torch.arange(0, t_rir.shape[-1], dtype=torch.int32)
    return torch.mean(torch.abs(db(t_edc[..., mask_index], is_squared=True) -
                                db(a_edc[..., mask_index], is_squared=True)))


def directional_edc_loss(H_pred: torch.Tensor, amps_true: torch.Tensor,
                         envelopes: torch.Tensor, mixing_time_samps: int,
                         edc_len_samps: int, mask_index: Optional[torch.Tensor] = None):
    """losses.py:333-371 -- irfft with default n = 2(K-1); envelopes (S,T) are an INPUT
    (slope2noise.decay_kernel is an absent dependency: parity unpinned for them)."""
    pred_rir = torch.fft.irfft(H_pred)[..., mixing_time_samps:edc_len_samps + mixing_time_samps]
    edc_pred = schroeder(pred_rir)
    edc_true = torch.einsum('bjk, kt -> bjt', amps_true.to(torch.float32), envelopes)
    if mask_index is None:
        mask_index = torch.arange(0, pred_rir.shape[-1], dtype=torch.int32)
    return torch.mean(torch.abs(db(edc_true[..., mask_index], is_squared=True) -
                                db(edc_pred[..., mask_index], is_squared=True)))


def sparsity_loss(A: torch.Tensor) -> torch.Tensor:
    """colorless_fdn/losses.py:7-17."""
    N = A.shape[-1]
    return -(torch.sum(torch.abs(A)) - (N * np.sqrt(N))) / (N * (np.sqrt(N) - 1))


def mse_loss(y_pred: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
    """colorless_fdn/losses.py:20-41 (1-D branch)."""
    return torch.mean(torch.pow(torch.abs(y_pred) - torch.abs(y_true), 2), dim=-1)


def amse_loss(y_pred: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
    """colorless_fdn/losses.py:44-73 (1-D branch): exponent 4 where |y|-|t| > 1 else 2."""
    gT = 2 * torch.ones(y_pred.shape, dtype=torch.float32)
    gT = gT + 2 * torch.gt(torch.abs(y_pred) - torch.abs(y_true), 1).type(torch.uint8)
    return torch.mean(torch.pow(torch.abs(y_pred) - torch.abs(y_true), gT), dim=0)


# --------------------------------------------------------------------------------------
# one full model evaluation + trainer step  (reference: trainer.py:259-332, 452-477)
# --------------------------------------------------------------------------------------
class GridModelParams:
    """Plain container of what DiffGFDNVarReceiverPos holds (state_dict surface, App. B)."""

    def __init__(self, sample_rate: float, delays: Sequence[int], num_groups: int,
                 input_gains: torch.Tensor, output_gains: torch.Tensor, M: torch.Tensor,
                 alpha: torch.Tensor, common_decay_times: np.ndarray,
                 mlp_weights, mlp_norms, num_fourier_features: int):
        self.sample_rate = sample_rate
        self.delays = torch.tensor(list(delays), dtype=torch.float32)
        self.num_groups = num_groups
        self.n_per_group = len(delays) // num_groups
        self.input_gains = input_gains
        self.output_gains = output_gains
        self.M = M
        self.alpha = alpha
        self.common_decay_times = common_decay_times
        self.mlp_weights = mlp_weights
        self.mlp_norms = mlp_norms
        self.num_fourier_features = num_fourier_features

    def gamma(self) -> torch.Tensor:
        """model.py:155-163 -- float64 numpy db2lin, flattened into a (float64) tensor.  ``common_decay_times`` given
        as a torch tensor = the learnable decay times of feedback_loop.py:205-232: the same expression as a
        differentiable function of T60 (one gain vector per group, concatenated)."""
        n = self.n_per_group
        if torch.is_tensor(self.common_decay_times):
            t60 = self.common_decay_times.reshape(-1)
            return torch.cat([torch.pow(10.0, (-60 * self.delays[i * n:(i + 1) * n] / (self.sample_rate * t60[i])) * 0.05)
                              for i in range(self.num_groups)])
        cdt = np.squeeze(self.common_decay_times)
        dl = self.delays.numpy().astype(np.int64)
        vals = [decay_times_to_gain_per_sample(cdt[i], dl[i * n:(i + 1) * n], self.sample_rate).tolist()
                for i in range(self.num_groups)]
        return torch.flatten(torch.tensor(vals))

    def receiver_gains(self, norm_pos: torch.Tensor) -> torch.Tensor:
        """gain_filters.py:497-524."""
        enc = sinusoidal_encoding(norm_pos, self.num_fourier_features)
        raw = mlp_forward(enc, self.mlp_weights, self.mlp_norms)
        return scaled_sigmoid(raw.view(-1), -1.0, 1.0).view(norm_pos.shape[0], self.num_groups)


def grid_model_forward(p: GridModelParams, batch: Dict[str, torch.Tensor],
                       use_colorless_loss: bool = True):
    """model.py:569-625."""
    z = batch['z_values']
    A = coupled_feedback_matrix(p.M, p.alpha)
    P = feedback_loop_forward(z, p.delays, p.gamma(), A)
    g = p.receiver_gains(batch['norm_listener_position'])
    H = var_receiver_forward(z, p.input_gains, p.output_gains, g, P,
                             batch['target_early_response'], p.n_per_group)
    if use_colorless_loss:
        return H, sub_fdn_output(z, p.M, p.input_gains, p.output_gains, p.delays)
    return H


def grid_losses(p: GridModelParams, batch, H, H_sub, *, edr_weight=1.0, edc_weight=1.0,
                spectral_weight=1.0, sparsity_weight=1.0, use_asym=False,
                subband_filter: Optional[torch.Tensor] = None,
                edc_mask: Optional[torch.Tensor] = None, mixing_time_ms: float = 20.0):
    """trainer.py:259-315 (omni branch) incl. the sparsity-overwrite quirk (:305-308)."""
    fs = p.sample_rate
    if subband_filter is not None:
        H = H * subband_filter                                              # trainer.py:459
    if p.common_decay_times is None:
        max_ms = 2000
    else:
        max_ms = float(np.max(p.common_decay_times)) * 1e3                  # trainer.py:56-59
    tgt = batch['target_rir_response']
    losses = {
        'edc_loss': edc_weight * edc_loss(tgt, H, ms_to_samps(max_ms, fs),
                                          ms_to_samps(mixing_time_ms, fs), edc_mask),
        'edr_loss': edr_weight * edr_loss(tgt, H),
    }
    if H_sub is not None:
        spec = 0.0
        spars = 0.0
        crit = amse_loss if use_asym else mse_loss
        for k in range(p.num_groups):
            hk = H_sub[0][..., k]
            spec = spec + spectral_weight * crit(hk, torch.ones_like(hk))
            spars = sparsity_weight * sparsity_loss(ortho_param(p.M[k]))   # overwritten!
        losses['spectral_loss'] = spec
        losses['sparsity_loss'] = spars
    return losses


def normalize_io_gains(p: GridModelParams, H_sub) -> None:
    """trainer.py:317-332 -- b_n, c_n /= (mean_k |Hout[k,g]|^2)^(1/4), in place."""
    with torch.no_grad():
        energy = torch.mean(torch.pow(torch.abs(H_sub[0]), 2), dim=0)
        n = p.n_per_group
        for prm in (p.input_gains, p.output_gains):
            for k in range(p.num_groups):
                prm.data[k * n:(k + 1) * n] /= torch.pow(energy[k], 1 / 4)
