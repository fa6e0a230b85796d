"""Feedback loop of the GFDN: orthogonal feedback-matrix parameterisation and the per-bin
resolvent (D_m(z) Gamma^{-1} - A)^{-1}.

Interface mirror of the reference's src/diff_gfdn/feedback_loop.py (FeedbackLoop :146-509,
constructor :148-160, forward :326-391).  Differences in HOW, not WHAT:
  * the reference materialises D, Gamma, A as (K, N, N) complex128 tensors and calls
    torch.linalg.inv; here the per-bin systems are SOLVED by the HIP kernel
    (csrc/solve.hip) against whichever right-hand side the caller needs
    (:meth:`resolvent_apply` -- the models use b, never the inverse);
  * with zero inter-group coupling A = blockdiag(Q_g Q_g) (reference :393-404 with Phi = I), so
    the kernel works on G independent n x n blocks instead of one N x N matrix;
  * ``forward(z)`` still returns the explicit (K, N, N) complex64 inverse for API parity (N solves).
The tiny parameterisation (expm of skew matrices, Givens rotations) stays on torch ops on the
device: G^2 products of 4x4..16x16 matrices per step.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from .config import CouplingMatrixType
from .functional import FrequencyGrid, OrthoParam, ResolventSolve, ResolventSolveFilter


class Skew(nn.Module):
    """X -> triu(X,1) - triu(X,1)^T   (reference :16-25)."""

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        A = X.triu(1)
        return A - A.transpose(-1, -2)


class MatrixExponential(nn.Module):
    """reference :28-36."""

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        return torch.matrix_exp(X)


class OrthoParamModule(nn.Sequential):
    """``ortho_param`` of the reference (:270: nn.Sequential(Skew(), MatrixExponential())) with the
    CUDA path routed to the HIP kernel, for one (n, n) matrix or a (G, n, n) stack."""

    def __init__(self):
        super().__init__(Skew(), MatrixExponential())

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        if X.is_cuda and X.dtype == torch.float32 and X.shape[-1] <= 32:
            M = X if X.ndim == 3 else X.unsqueeze(0)
            Q = OrthoParam.apply(M.contiguous())[0]
            return Q if X.ndim == 3 else Q[0]
        return super().forward(X)


class ND_Unitary(nn.Module):
    """N x N rotation from N(N-1)/2 Givens angles, U_n = R_{n-2}..R_0 diag(U_{n-1}, 1)
    (reference :39-87).  N == 1 returns the python int 1, as the reference does."""

    def forward(self, alpha: torch.Tensor, N: int):
        assert len(alpha) == N * (N - 1) // 2
        if N == 1:
            return 1
        eye = torch.eye(N, dtype=alpha.dtype, device=alpha.device)
        start = (N - 1) * (N - 2) // 2
        cur = alpha[start:]
        rot = eye
        for i in range(N - 1):
            c, s = torch.cos(cur[i]), torch.sin(cur[i])
            R = eye.clone()
            R[i, i] = c
            R[i, -1] = -s
            R[-1, i] = s
            R[-1, -1] = c
            rot = R @ rot
        big = eye.clone()
        big[:N - 1, :N - 1] = self.forward(alpha[:start], N - 1)
        return rot @ big


def matrix_convolution(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """Product of two polynomial matrices (M, N, K) x (N, Q, R) -> (M, Q, K + R - 1): C[m, q] = sum_n A[m, n] * B[n, q]
    with * the full linear convolution along the last axis (utils.py:216-239).  One einsum per tap of the shorter
    factor, out of place, so that autograd sees ordinary tensor ops."""
    K, R = A.shape[-1], B.shape[-1]
    if K <= R:
        terms = [nn.functional.pad(torch.einsum('mn,nqr->mqr', A[..., i], B), (i, K - 1 - i)) for i in range(K)]
    else:
        terms = [nn.functional.pad(torch.einsum('mnk,nq->mqk', A, B[..., j]), (j, R - 1 - j)) for j in range(R)]
    return torch.stack(terms).sum(0)


class FIRParaunitary(nn.Module):
    """FIR paraunitary matrix (N, N, order) as a cascade of order-1 Householder factors
    I - (1 - z^-1) v v^T times a unitary matrix (reference :90-143)."""

    def __init__(self, N: int, order: int):
        super().__init__()
        self.N = N
        self.order = order

    def construct_elementary_householder_matrix(self, unit_vector: torch.Tensor) -> torch.Tensor:
        assert len(unit_vector) == self.N
        vv = torch.outer(unit_vector, unit_vector)
        return torch.stack([torch.eye(self.N, dtype=vv.dtype, device=vv.device) - vv, vv], dim=-1)

    def forward(self, unitary_matrix: torch.Tensor, unit_vectors: torch.Tensor) -> torch.Tensor:
        assert unitary_matrix.shape == (self.N, self.N)
        assert unit_vectors.shape == (self.N, self.order - 1)
        poly = torch.eye(self.N, dtype=unit_vectors.dtype, device=unit_vectors.device)[..., None]
        for k in range(self.order - 1):
            poly = matrix_convolution(self.construct_elementary_householder_matrix(unit_vectors[:, k]), poly)
        return matrix_convolution(poly, unitary_matrix.reshape(self.N, self.N, 1))


def decay_times_to_gain_per_sample(common_decay_times, delay_length_samp, fs):
    """gamma = 10^(0.05 * (-60 m / (fs T60)))   (reference absorption_filters.py:40-53)."""
    if isinstance(common_decay_times, torch.Tensor):
        return torch.pow(10.0, (-60 * delay_length_samp / (fs * common_decay_times)) * 0.05)
    return np.power(10.0, (-60 * np.array(delay_length_samp) / (fs * common_decay_times)) * 0.05)


class FeedbackLoop(nn.Module):

    def __init__(self,
                 sample_rate: float,
                 num_groups: int,
                 num_delay_lines_per_group: int,
                 delays: torch.Tensor,
                 use_absorption_filters: bool,
                 coupling_matrix_type: CouplingMatrixType = None,
                 use_zero_coupling: bool = True,
                 coupling_matrix_order: Optional[int] = None,
                 colorless_feedback_matrix: Optional[torch.Tensor] = None,
                 gains: Optional[torch.Tensor] = None,
                 common_decay_times: Optional[List] = None,
                 device: torch.device = 'cpu', precise_solve: bool = False):
        super().__init__()
        # float64 per-bin systems (the reference always inverts in complex128; float32 is 1e-4-exact at the decay
        # times of the room models, not for nearly lossless loops with T60 of tens of seconds)
        self.precise_solve = precise_solve
        if use_absorption_filters and gains is None:
            raise NotImplementedError("absorption filters are fixed designs (reference :200-201 'Cannot learn "
                                      "absorption filters yet'): pass their coefficients as ``gains``")
        self.sample_rate = sample_rate
        self.num_groups = num_groups
        self.num_delay_lines_per_group = num_delay_lines_per_group
        # non-persistent buffers: follow .to(device) without adding state-dict keys
        self.register_buffer('delays', torch.as_tensor(delays, dtype=torch.float32).clone(),
                             persistent=False)
        self.num_delays = len(self.delays)
        self.use_absorption_filters = use_absorption_filters
        self.use_zero_coupling = use_zero_coupling
        self.device = device
        self.coupling_matrix_type = coupling_matrix_type
        self.coupling_matrix_order = coupling_matrix_order
        self.ortho_param = OrthoParamModule()
        self._ortho_cache = None
        self._inv_gamma_cache = None
        self._zp_cache = None
        self._eps = 1e-9
        self._init_absorption(gains, common_decay_times)
        self._init_feedback_matrix(colorless_feedback_matrix)

    # -- reference :193-258 (scalar-gain branches) ------------------------------------------------
    def _init_absorption(self, gains, common_decay_times):
        G, n = self.num_groups, self.num_delay_lines_per_group
        self.learn_decay_times = gains is None
        if gains is None:
            if common_decay_times is None:
                init = 0.1 + (2.0 - 0.1) * torch.rand(G)
            else:
                init = torch.tensor(np.asarray(common_decay_times).squeeze())
            self.common_decay_times = nn.Parameter(init)
            self.delays_by_group = [self.delays[i:i + n] for i in range(0, self.num_delays, n)]
        elif self.use_absorption_filters:
            # (N, S, 3, 2) second-order sections [..., 0] numerator / [..., 1] denominator (GEQ design,
            # reference :246-255) or (N, order, 2) direct-form IIR (Prony design, :238-243): coefficient DATA;
            # the responses Gamma_i(z_k) are evaluated once per frequency grid (absorption_response)
            self.delay_line_gains = torch.as_tensor(gains)
            if self.delay_line_gains.ndim not in (3, 4) or self.delay_line_gains.shape[-1] != 2:
                raise ValueError("absorption filter coefficients: (N, S, 3, 2) SOS or (N, order, 2) IIR")
            self._abs_cache = None
        else:
            # plain attribute like the reference (:258); moved by _apply below, and re-linked to
            # the owning model's persistent ``delay_filters`` buffer by DiffGFDN._apply
            self.delay_line_gains = torch.as_tensor(gains)

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        if not self.learn_decay_times:
            self.delay_line_gains = fn(self.delay_line_gains)
            if self.use_absorption_filters:
                self._abs_cache = None
        return self

    def absorption_response(self, z: torch.Tensor) -> torch.Tensor:
        """Gamma_i(z_k) (N, K) complex64 of the per-line absorption filters: cascades of second-order sections
        prod_s (b0 + b1 z^-1 + b2 z^-2) / (a0 + a1 z^-1 + a2 z^-2) (gain_filters.py:221-241, accumulated in
        complex64 like the reference) or direct-form IIR with the reference's +1e-9 in the denominator
        (gain_filters.py:180-198)."""
        g = self.delay_line_gains.to(z.device)
        z = z.to(torch.complex128)
        zi = 1.0 / z
        if g.ndim == 4:
            H = torch.ones((g.shape[0], z.numel()), dtype=torch.complex64, device=z.device)
            for s_ in range(g.shape[1]):
                b, a = g[:, s_, :, 0].to(torch.float64), g[:, s_, :, 1].to(torch.float64)
                num = b[:, 0:1] + b[:, 1:2] * zi[None, :] + b[:, 2:3] * zi[None, :] ** 2
                den = a[:, 0:1] + a[:, 1:2] * zi[None, :] + a[:, 2:3] * zi[None, :] ** 2
                H = (H * (num / den)).to(torch.complex64)
            return H
        num = torch.zeros((g.shape[0], z.numel()), dtype=torch.complex64, device=z.device)
        den = torch.zeros_like(num)
        for k in range(g.shape[1]):
            zk = torch.pow(z, -k)
            num = (num + g[:, k, 0].to(torch.float64)[:, None] * zk[None, :]).to(torch.complex64)
            den = (den + g[:, k, 1].to(torch.float64)[:, None] * zk[None, :]).to(torch.complex64)
        return num / (den + 1e-9)

    def _inv_gamma_bins(self, z: torch.Tensor) -> torch.Tensor:
        """(K, N) complex64 = 1 / Gamma_i(z_k), cached per frequency grid (the filters are fixed)."""
        key = (z.data_ptr(), z.numel(), z._version, str(z.device))
        if self._abs_cache is None or self._abs_cache[0] != key:
            G = self.absorption_response(z)
            self._abs_cache = (key, (1.0 / G.to(torch.complex128)).to(torch.complex64).T.contiguous(), z)
        return self._abs_cache[1]

    def current_gains(self) -> torch.Tensor:
        """gamma_n (N,), differentiable w.r.t. learnable decay times (reference :221-232)."""
        if self.learn_decay_times:
            dev = self.common_decay_times.device
            return torch.cat([decay_times_to_gain_per_sample(
                self.common_decay_times[i], self.delays_by_group[i].to(dev),
                torch.tensor(self.sample_rate, device=dev)) for i in range(self.num_groups)])
        return self.delay_line_gains

    # -- reference :260-324 ------------------------------------------------------------------------
    def _init_feedback_matrix(self, colorless_feedback_matrix):
        G, n = self.num_groups, self.num_delay_lines_per_group
        if self.coupling_matrix_type == CouplingMatrixType.RANDOM:
            self.random_feedback_matrix = nn.Parameter(
                (2 * torch.rand(self.num_delays, self.num_delays) - 1) / np.sqrt(n))
            return
        if colorless_feedback_matrix is not None:
            self.register_buffer('M', colorless_feedback_matrix.clone().detach(), persistent=False)
        else:
            self.M = nn.Parameter((2 * torch.rand(G, n, n) - 1) / np.sqrt(n))
        if self.coupling_matrix_type == CouplingMatrixType.FILTER:
            # order - 1 Householder vectors and the zeroth-order unitary factor (reference :311-323)
            self.unit_vectors = nn.Parameter(torch.randn(G, self.coupling_matrix_order - 1))
            self.unitary_matrix = nn.Parameter((2 * torch.rand(G, G) - 1) / np.sqrt(G))
            self.fir_paraunitary = FIRParaunitary(G, self.coupling_matrix_order)
            return
        self.nd_unitary = ND_Unitary()
        if self.use_zero_coupling:
            self.register_buffer("alpha", torch.zeros(G * (G - 1) // 2))
        else:
            self.alpha = nn.Parameter(np.pi / 4 * torch.rand(G * (G - 1) // 2, dtype=torch.float32))

    # -- reference :393-455 ------------------------------------------------------------------------
    def _ortho(self):
        """(Q, QQ) of the current parameters, shared by the solve and the sparsity loss of ONE
        forward: callers open a scope with :meth:`new_forward` (models and the trainer step do),
        inside which the pair is evaluated once.  The key (version counter, grad mode) only guards
        against misuse inside a scope."""
        M = self.M
        key = (M.data_ptr(), M._version, torch.is_grad_enabled() and M.requires_grad)
        if self._ortho_cache is None or self._ortho_cache[0] != key:
            self._ortho_cache = (key, OrthoParam.apply(M))
        return self._ortho_cache[1]

    def new_forward(self):
        """Drop per-forward caches (parameters may have been updated in place by a fused kernel)."""
        self._ortho_cache = None

    def group_rotations(self) -> torch.Tensor:
        """Q_g = expm(skew(M_g)), (G, n, n) -- HIP kernel on the device (csrc/ortho.hip)."""
        if self.M.is_cuda:
            return self._ortho()[0]
        return self.ortho_param(self.M)

    def construct_block_mixing_matrix(self) -> torch.Tensor:
        Q = self.group_rotations()
        G, n = self.num_groups, self.num_delay_lines_per_group
        blocks = torch.einsum('iab,jbc->iajc', Q, Q)          # block (i,j) = Q_i Q_j
        return blocks.reshape(G * n, G * n)

    @property
    def uncoupled(self) -> bool:
        """A = blockdiag(Q_g Q_g): SCALAR coupling with the angles fixed at zero (reference :296-304)."""
        return (self.coupling_matrix_type not in (CouplingMatrixType.RANDOM, CouplingMatrixType.FILTER)
                and self.use_zero_coupling)

    def construct_coupling_matrix(self):
        if self.coupling_matrix_type == CouplingMatrixType.FILTER:
            # (G, G, order) paraunitary FIR matrix from unit-norm Householder vectors (reference :413-420)
            v = self.unit_vectors / (torch.norm(self.unit_vectors, dim=0, keepdim=True) + self._eps)
            return self.fir_paraunitary(self.ortho_param(self.unitary_matrix), v)
        alpha = self.alpha.clamp(min=-np.pi, max=np.pi)
        return self.nd_unitary(alpha, self.num_groups)

    def get_coupled_feedback_matrix(self) -> torch.Tensor:
        """A = block_M o kron(Phi, 1) as complex64 (N, N), or (N, N, order) with A[..., p] = block_M o kron(Phi_p, 1)
        for FILTER coupling   (reference :424-455)."""
        if self.coupling_matrix_type == CouplingMatrixType.FILTER:
            block_M = self.construct_block_mixing_matrix()
            self.phi = self.construct_coupling_matrix()
            n = self.num_delay_lines_per_group
            A = block_M[..., None] * self.phi.repeat_interleave(n, 0).repeat_interleave(n, 1)
            return torch.complex(A, torch.zeros_like(A))
        A = self._real_feedback_matrix()
        return torch.complex(A, torch.zeros_like(A))

    def _real_feedback_matrix(self) -> torch.Tensor:
        if self.coupling_matrix_type == CouplingMatrixType.RANDOM:
            return self.ortho_param(self.random_feedback_matrix)
        block_M = self.construct_block_mixing_matrix()
        self.phi = self.construct_coupling_matrix()
        if self.num_groups == 1:
            return block_M                       # reference falls back to block_M (:441-445)
        n = self.num_delay_lines_per_group
        ones = torch.ones((n, n), dtype=block_M.dtype, device=block_M.device)
        return block_M * torch.kron(self.phi, ones)

    def feedback_blocks(self) -> torch.Tensor:
        """What the solver consumes: (G, n, n) diagonal blocks Q_g Q_g when the groups are
        uncoupled, else the dense (1, N, N) matrix."""
        if self.uncoupled:
            if self.M.is_cuda:
                return self._ortho()[1]
            Q = self.group_rotations()
            return Q @ Q
        return self._real_feedback_matrix().unsqueeze(0)

    # -- the hot path ------------------------------------------------------------------------------
    def resolvent_apply(self, z: torch.Tensor, b: torch.Tensor,
                        transpose: bool = False) -> torch.Tensor:
        """Y[k] = (D(z_k) Gamma^{-1} - A)^{-1} b  -> (K, N) complex64; A^T when ``transpose``."""
        grid = FrequencyGrid.of(z)
        if self.coupling_matrix_type == CouplingMatrixType.FILTER:
            return self._resolvent_apply_filter(z, grid, b, transpose)
        A = self.feedback_blocks()
        dev = A.device
        if self.use_absorption_filters:
            if getattr(self, '_ones_n', None) is None or self._ones_n.device != dev:
                self._ones_n = torch.ones(self.num_delays, dtype=torch.float32, device=dev)
            return ResolventSolve.apply(A, self._ones_n, b.reshape(-1), grid, self.delays, transpose,
                                        self._inv_gamma_bins(z))
        if self.precise_solve:
            # float64 inverse gains: at pole radii ~0.9999 the float32 rounding of gamma alone moves the peaks
            inv_gamma = 1.0 / self.current_gains().to(dev).to(torch.float64)
        elif self.learn_decay_times:
            inv_gamma = (1.0 / self.current_gains().to(dev)).to(torch.float32)
        else:
            g = self.delay_line_gains
            key = (g.data_ptr(), g._version, str(dev))
            if self._inv_gamma_cache is None or self._inv_gamma_cache[0] != key:
                self._inv_gamma_cache = (key, (1.0 / g.to(dev)).to(torch.float32))
            inv_gamma = self._inv_gamma_cache[1]
        return ResolventSolve.apply(A, inv_gamma, b.reshape(-1), grid, self.delays, transpose, None,
                                    self.precise_solve)

    def coupling_response(self, z: torch.Tensor, phi: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Phi(z_k) = sum_p Phi_p z_k^-p -> (K, G, G) complex64 (the frequency dependence of reference :362-373);
        the powers z_k^-p are tabulated once per grid in complex128."""
        key = (z.data_ptr(), z.numel(), z._version, str(z.device))
        if self._zp_cache is None or self._zp_cache[0] != key:
            p = torch.arange(self.coupling_matrix_order, device=z.device, dtype=torch.float64)
            zp = torch.exp(-p[None, :] * torch.log(z.to(torch.complex128))[:, None])
            self._zp_cache = (key, zp.to(torch.complex64), z)
        phi = self.construct_coupling_matrix() if phi is None else phi
        return torch.einsum('ghp,kp->kgh', phi.to(torch.complex64), self._zp_cache[1])

    def _resolvent_apply_filter(self, z, grid, b, transpose):
        BM = self.construct_block_mixing_matrix()
        self.phi = self.construct_coupling_matrix()
        Phi = self.coupling_response(z, self.phi)
        if transpose:
            BM, Phi = BM.T, Phi.transpose(1, 2)
        dev = BM.device
        if self.use_absorption_filters:
            # absorption FILTERS on the lines as well (reference :332-344, :376-381 in the same pass as :362-373): the
            # per-bin complex inverse responses ride the diagonal, the real factor is one
            if getattr(self, '_ones_n', None) is None or self._ones_n.device != dev:
                self._ones_n = torch.ones(self.num_delays, dtype=torch.float32, device=dev)
            return ResolventSolveFilter.apply(BM.contiguous(), Phi.contiguous(), self._ones_n, b.reshape(-1), grid,
                                              self.delays, self.num_delay_lines_per_group, self._inv_gamma_bins(z))
        inv_gamma = (1.0 / self.current_gains().to(dev)).to(torch.float32)
        return ResolventSolveFilter.apply(BM.contiguous(), Phi.contiguous(), inv_gamma, b.reshape(-1), grid,
                                          self.delays, self.num_delay_lines_per_group)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        """Explicit (K, N, N) complex64 inverse, for API parity with reference :326-391."""
        N = self.num_delays
        dev = z.device
        cols = [self.resolvent_apply(z, torch.eye(N, device=dev)[j]) for j in range(N)]
        self.coupled_feedback_matrix = self.get_coupled_feedback_matrix()
        return torch.stack(cols, dim=-1)

    @torch.no_grad()
    def get_param_dict(self) -> Dict:
        out = {'delay_line_gains': self.current_gains()}
        if hasattr(self, 'common_decay_times'):
            out['common_decay_times'] = self.common_decay_times
        A = self.get_coupled_feedback_matrix()
        out['coupled_feedback_matrix'] = A.squeeze().cpu().numpy()
        if self.coupling_matrix_type != CouplingMatrixType.RANDOM:
            if not self.use_zero_coupling and hasattr(self, 'alpha'):
                out['coupling_coefficient'] = self.alpha.squeeze().cpu().numpy()
            out['coupling_matrix'] = (self.phi.squeeze().cpu().numpy()
                                      if torch.is_tensor(self.phi) else np.asarray(self.phi))
            out['individual_mixing_matrix'] = self.M.squeeze().cpu().numpy()
        return out
