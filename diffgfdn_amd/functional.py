"""torch.autograd Functions over the HIP kernels (diffgfdn_amd.hip_ops).

Each Function is one differentiable stage of the hot path; forward and backward both run
hand-written kernels, torch only carries the tensors.  Gradient conventions are PyTorch's
(complex grad = dL/dRe + i dL/dIm).
"""
import contextlib
from typing import Optional

import torch

from . import hip_ops as ops


class FrequencyGrid:
    """(turns, logr) of a z grid, computed once per distinct ``z_values`` tensor.

    reference: dataloader.py:552-566 builds z = polar(r, 2 pi rfftfreq(nfft)); the models raise
    it to the delay lengths every forward (feedback_loop.py:330)."""

    _cache = {}
    _capacity = 64

    def __init__(self, z: torch.Tensor):
        # (the cache key is the tensor's ADDRESS and version: the entry keeps the tensor alive, so that the allocator cannot
        # hand its block to another z of the same length while the entry exists -- a freed grid's address reused by a
        # different grid was a false hit waiting to happen: wrong phasors, H off by 4e-3 in one test order)
        self.z = z
        self.K = z.numel()
        self.turns, logr = ops.zprep(z)
        # the unit circle is by far the common case: skip the radius factor entirely
        self.on_unit_circle = bool((logr.abs().max() < 1e-12).item())
        self.logr = None if self.on_unit_circle else logr
        # uniform grids on the unit circle (the reference's z = exp(2 pi i rfftfreq(nfft)), dataloader.py:552-566):
        # the step in turns, else 0 -- kernels that walk consecutive bins may then rotate instead of re-evaluating
        self.dturn = 0.0
        if self.on_unit_circle and self.K > 2:
            d = float(((self.turns[-1] - self.turns[0]) / (self.K - 1)).item())
            k = torch.arange(self.K, dtype=torch.float64, device=self.turns.device)
            if d != 0.0 and bool(((self.turns - (self.turns[0] + k * d)).abs().max() < 1e-13).item()):
                self.dturn = d
        # the reference's own grid, bins 0 .. nfft / 2 of an nfft-point transform with nfft a power of two: z_k^m is a root of
        # unity of order nfft for integer m -- polynomials in z with integer degrees are then real transforms of their
        # coefficient sequences (csrc/polyfft.hip).  0: any other grid
        self.rfft_nfft = 0
        nfft = 2 * (self.K - 1)
        if (self.dturn != 0.0 and nfft >= 16 and nfft & (nfft - 1) == 0 and abs(self.dturn * nfft - 1.0) < 1e-12
                and float(self.turns[0].item()) == 0.0):
            self.rfft_nfft = nfft

    @classmethod
    def of(cls, z: torch.Tensor) -> "FrequencyGrid":
        key = (z.data_ptr(), z.numel(), z._version, str(z.device))
        g = cls._cache.get(key)
        if g is None:
            if torch.cuda.is_current_stream_capturing():
                # (building a grid reads the device: a step must have met its z BEFORE it is captured -- the warm-up does)
                raise RuntimeError("FrequencyGrid: first use of this z_values tensor inside a stream capture; run the step "
                                   "once before capturing it")
            # Oldest entry out, one at a time.  (The cache used to be CLEARED at nine entries: seven bands' steps, each with
            # its own copy of z, in a process whose cache already held a few grids lost the first bands' entries during the
            # later bands' warm-up -- and then built them again, with their host reads, inside the capture:
            # hipErrorStreamCaptureUnsupported / ...Unjoined.)  Steps keep the grids they recorded pointers of alive
            # themselves (GraphedModuleStep._grids).
            while len(cls._cache) >= cls._capacity:
                cls._cache.pop(next(iter(cls._cache)))
            g = cls(z)
            cls._cache[key] = g
        return g


class OrthoParam(torch.autograd.Function):
    """(Q, QQ) = (expm(skew(M)), expm(skew(M))^2) for M (G, n, n)  (feedback_loop.py:16-36, :270,
    :393-404), forward and adjoint in one HIP kernel each."""

    @staticmethod
    def forward(ctx, M):
        Q, QQ = ops.ortho_fwd(M, True, True)
        ctx.save_for_backward(M, Q)
        return Q, QQ

    @staticmethod
    def backward(ctx, gQ, gQQ):
        M, Q = ctx.saved_tensors
        return ops.ortho_bwd(M, gQ.contiguous(), gQQ.contiguous(), Q).to(M.dtype)


class RowNormalise(torch.autograd.Function):
    """w / (||w||_2 + 1e-6) over the last axis (reference spatial_sampling/model.py:117-190 ``normalise_weights``): one
    launch each way instead of 3 + 11 tensor operators."""

    @staticmethod
    def forward(ctx, w):
        ctx.save_for_backward(w)
        return ops.rownorm_fwd(w, 1e-6)

    @staticmethod
    def backward(ctx, gy):
        (w,) = ctx.saved_tensors
        return ops.rownorm_bwd(w, gy.contiguous(), 1e-6)


class MlpGains(torch.autograd.Function):
    """gains (B, G) = ScaledSigmoid(MLP(SinusoidalEncoding(pos)))  (gain_filters.py:497-524) in one
    launch; parameters are passed as separate tensors (autograd hands each its gradient) and packed
    into one flat buffer with a single cat -- or with no copy at all when they already are
    consecutive views of one buffer (FlatAdam's parameter buffer).  ``rows``: optional int64 index,
    item b uses pos[rows[b]] (pos = positions of ALL receivers)."""

    @staticmethod
    def _packed(params):
        p0 = params[0]
        off = 0
        for p in params:
            if (not p.is_contiguous() or p.dtype != torch.float32
                    or p.data_ptr() != p0.data_ptr() + 4 * off
                    or p.untyped_storage().data_ptr() != p0.untyped_storage().data_ptr()):
                return torch.cat([q.reshape(-1) for q in params])
            off += p.numel()
        return torch.as_strided(p0.detach(), (off,), (1,), p0.storage_offset())

    @staticmethod
    def forward(ctx, pos, rows, freq_pi, H, n_hidden, G, lo, hi, *params):
        # band-stacked call (BandBank): ONE parameter tensor (nbands, P), each row a packed set
        ctx.nbands = params[0].shape[0] if len(params) == 1 and params[0].dim() == 2 else 1
        if isinstance(H, (tuple, list)):          # (bands with their own layer sizes: ONE packed vector, band after band)
            ctx.nbands = len(H)
        w = params[0].detach() if ctx.nbands > 1 or len(params) == 1 else MlpGains._packed(params)
        gains, xhat, rstd = ops.mlp_gains_fwd(pos, freq_pi, w, H, n_hidden, G, lo, hi, rows, ctx.nbands)
        ctx.rows = rows
        ctx.save_for_backward(pos, freq_pi, w, gains, xhat, rstd)
        ctx.cfg = (H, n_hidden, G, lo, hi)
        ctx.shapes = [p.shape for p in params]
        return gains

    @staticmethod
    def backward(ctx, ggains):
        pos, freq_pi, w, gains, xhat, rstd = ctx.saved_tensors
        H, n_hidden, G, lo, hi = ctx.cfg
        gw = ops.mlp_gains_bwd(pos, freq_pi, w, H, n_hidden, G, lo, hi, gains, xhat, rstd,
                               ggains.contiguous(), ctx.rows, ctx.nbands)
        if len(ctx.shapes) == 1:
            return (None,) * 8 + (gw.view(ctx.shapes[0]),)
        grads, off = [], 0
        for shp in ctx.shapes:
            n = 1
            for v in shp:
                n *= v
            grads.append(gw[off:off + n].view(shp))
            off += n
        return (None, None, None, None, None, None, None, None) + tuple(grads)


class ResolventSolve(torch.autograd.Function):
    """Y[k] = (diag(z_k^m inv_gamma) - A)^{-1} b   (A^T when transpose)."""

    @staticmethod
    def forward(ctx, A, inv_gamma, b, grid: FrequencyGrid, delays, transpose: bool, inv_gamma_bins=None,
                precise: bool = False):
        """``inv_gamma_bins`` (K, N) complex64: fixed frequency-dependent absorption 1 / Gamma_i(z_k)
        (feedback_loop.py:332-344); ``inv_gamma`` is then a vector of ones.  ``precise``: float64 matrix entries
        and elimination (the reference inverts in complex128, feedback_loop.py:389-391)."""
        Y = ops.solve_fwd(grid.turns, grid.logr, A, delays, inv_gamma, b, transpose, inv_gamma_bins, precise)
        ctx.precise = precise
        # Y is kept alive by its consumer's backward anyway (the output stage reads it): saving
        # it here costs nothing and the backward kernel skips re-solving the forward system
        ctx.save_for_backward(A, inv_gamma, b, delays, Y, inv_gamma_bins)
        ctx.grid = grid
        ctx.transpose = transpose
        return Y

    @staticmethod
    def backward(ctx, gY):
        A, inv_gamma, b, delays, Y, igz = ctx.saved_tensors
        g = ctx.grid
        gA, gb, gig = ops.solve_bwd(g.turns, g.logr, A, delays, inv_gamma, b, gY.contiguous(),
                                    ctx.transpose, Y=Y, inv_gamma_bins=igz, precise=ctx.precise)
        return (gA.to(A.dtype), None if gig is None else gig.to(inv_gamma.dtype), gb.to(b.dtype).reshape(b.shape),
                None, None, None, None, None)


class ResolventSolveFilter(torch.autograd.Function):
    """Y[k] = (diag(z_k^m inv_gamma) - BM o kron(Phi_k, 1))^{-1} b: paraunitary FILTER coupling
    (feedback_loop.py:362-373, :441-455), Phi (K, G, G) complex64 per bin."""

    @staticmethod
    def forward(ctx, BM, Phi, inv_gamma, b, grid: FrequencyGrid, delays, nper: int, inv_gamma_bins=None):
        # ``inv_gamma_bins`` (K, N) complex64: 1 / Gamma_i(z_k) of fixed absorption filters (feedback_loop.py:332-344,
        # :376-381) -- data, no gradient
        Y = ops.solve_phi_fwd(grid.turns, grid.logr, BM, Phi, nper, delays, inv_gamma, b, inv_gamma_bins)
        ctx.save_for_backward(BM, Phi, inv_gamma, b, delays, Y)
        ctx.grid, ctx.nper, ctx.igb = grid, nper, inv_gamma_bins
        return Y

    @staticmethod
    def backward(ctx, gY):
        BM, Phi, inv_gamma, b, delays, Y = ctx.saved_tensors
        g = ctx.grid
        gBM, gb, gig, gPhi = ops.solve_phi_bwd(g.turns, g.logr, BM, Phi, ctx.nper, delays, inv_gamma, b,
                                               gY.contiguous(), Y, ctx.igb)
        return (gBM.to(BM.dtype), gPhi, gig.to(inv_gamma.dtype), gb.to(b.dtype).reshape(b.shape), None, None, None,
                None)


class SvfCoefficients(torch.autograd.Function):
    """Unconstrained SVF parameters (..., S, 2) -> biquad coefficients (..., S, 6) in one launch each way
    (gain_filters.py:327-330, :36-103, :117-151)."""

    @staticmethod
    def forward(ctx, raw, cutoff, compress_pole_factor: float):
        ctx.save_for_backward(raw, cutoff)
        ctx.cpf = compress_pole_factor
        return ops.svf_coefficients(raw, cutoff, compress_pole_factor)

    @staticmethod
    def backward(ctx, gcoef):
        raw, cutoff = ctx.saved_tensors
        return ops.svf_coefficients(raw, cutoff, ctx.cpf, gcoef.contiguous()).to(raw.dtype), None, None


class SosOutputStage(torch.autograd.Function):
    """H[b][k] = sum_g cascade_{b,g}(z_k) T[k][g] + direct[b][k]: second-order-section cascades (SVF output filters,
    gain_filters.py:221-241, :262-402) contracted with the group transfer functions (model.py:588-619) in one
    kernel; coef (B, G, S, 6) float32 biquad coefficients, T (K, G) complex64."""

    @staticmethod
    def forward(ctx, coef, T, direct, z):
        H = ops.sos_compose_fwd(coef, z, T, direct)
        ctx.save_for_backward(coef, T, z)
        return H

    @staticmethod
    def backward(ctx, gH):
        coef, T, z = ctx.saved_tensors
        gcoef, gT = ops.sos_compose_bwd(coef, z, T, gH.contiguous())
        return gcoef.to(coef.dtype), gT.to(T.dtype), None, None


class OutputStage(torch.autograd.Function):
    """H[b][k] = (sum_g rgain[b][g] sum_{n in g} c_n Y[k][n] + direct[b][k]) * filt[k]."""

    @staticmethod
    def forward(ctx, Y, c, rgain, nper: int, direct, filt, direct_rows=None, nbands: int = 1):
        H = ops.compose_fwd(Y, c, rgain, nper, direct, filt, direct_rows=direct_rows, nbands=nbands)
        ctx.save_for_backward(Y, c, rgain, filt)
        ctx.nper, ctx.nbands = nper, nbands
        return H

    @staticmethod
    def backward(ctx, gH):
        Y, c, rgain, filt = ctx.saved_tensors
        gY, gc, grg = ops.compose_bwd(Y, c, rgain, ctx.nper, gH.contiguous(), filt, ctx.nbands)
        return gY, gc.to(c.dtype).reshape(c.shape), grg.to(rgain.dtype), None, None, None, None, None


class GroupSums(torch.autograd.Function):
    """S (G, K) = sum_{n in g} c_n Y[k][n]: the sub-FDN group responses (model.py:243-250) without the receiver machinery
    of the general output stage."""

    @staticmethod
    def forward(ctx, Y, c, G: int, nper: int):
        ctx.save_for_backward(Y, c)
        ctx.G, ctx.nper = G, nper
        return ops.group_sums_fwd(Y, c, G, nper)

    @staticmethod
    def backward(ctx, gS):
        Y, c = ctx.saved_tensors
        gY, gc = ops.group_sums_bwd(Y, c, ctx.G, ctx.nper, gS.contiguous())
        return gY, gc.to(c.dtype).reshape(c.shape), None, None


class SubFdnTransforms(torch.autograd.Function):
    """S (G, K) = c_g^T (D(z) - M_g)^{-1} b_g, the un-damped group responses of the colorless branch (model.py:209-252), for
    zero-coupling blocks of up to nine lines with INTEGER delay lengths on the reference's rfftfreq grid: numerator and
    determinant polynomials as real transforms of their coefficient sequences, the adjoint as a gather from two inverse
    transforms and a float64 cofactor map (csrc/blocktf9.hip, csrc/polyfft.hip) -- in place of the per-bin n x n complex
    eliminations of ``ResolventSolve`` + ``GroupSums``, which hold the whole chip for their two launches."""

    @staticmethod
    def forward(ctx, M, b, c, delays, nper: int, nfft: int, T_seq: int):
        coef = ops.tf9_coefs(M, None, b)
        X = ops.tfp_forward(coef, delays, c, nper, nfft, T_seq)
        nblk = M.shape[0]
        S, Dinv = ops.tfp_ratio_fwd(X[:nblk], X[nblk:])
        ctx.save_for_backward(M, b, c, delays, S, Dinv)
        ctx.meta = (nper, nfft, T_seq)
        return S

    @staticmethod
    def backward(ctx, gS):
        M, b, c, delays, S, Dinv = ctx.saved_tensors
        nper, nfft, T_seq = ctx.meta
        part = ops.tfp_ratio_bwd(nfft, nper, delays, gS, S, Dinv, T_seq=T_seq)
        gA, gb, gc = ops.tf9_rec_grads(M, None, part, b, c)
        return gA.to(M.dtype).reshape(M.shape), gb.to(b.dtype).reshape(b.shape), gc.to(c.dtype).reshape(c.shape), None, None, \
            None, None


class SHOutputStage(torch.autograd.Function):
    """H_sh[b][l][k] = filt[k] * sum_g w[b][g][l] c_{g,l} Y[k][g nper + l]  (model.py:1056-1088)."""

    @staticmethod
    def forward(ctx, Y, c, w, G: int, nper: int, filt):
        H = ops.compose_sh_fwd(Y, c, w, G, nper, filt)
        ctx.save_for_backward(Y, c, w, filt)
        ctx.G, ctx.nper = G, nper
        return H

    @staticmethod
    def backward(ctx, gH):
        Y, c, w, filt = ctx.saved_tensors
        gY, gc, gw = ops.compose_sh_bwd(Y, c, w, ctx.G, ctx.nper, gH.contiguous(), filt)
        return gY, gc.to(c.dtype).reshape(c.shape), gw.to(w.dtype).reshape(w.shape), None, None, None


class SHToDirectional(torch.autograd.Function):
    """H_dir = einsum('jl,blk->bjk', A, H_sh)  (trainer.py:853-865)."""

    @staticmethod
    def forward(ctx, A, H_sh):
        ctx.save_for_backward(A)
        return ops.sh_to_directional(A, H_sh, False)

    @staticmethod
    def backward(ctx, gH):
        (A,) = ctx.saved_tensors
        return None, ops.sh_to_directional(A, gH.contiguous(), True)


class SpectralLoss(torch.autograd.Function):
    """sum_g mean_k (|S[g][k]| - 1)^p  (colorless_fdn/losses.py:20-73); gradient fused."""

    @staticmethod
    def forward(ctx, S, asym: bool):
        energy, loss, gS = ops.spectral_stats(S, asym, 1.0, want_grad=S.requires_grad)
        ctx.save_for_backward(gS)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (gS,) = ctx.saved_tensors
        return gS * gloss.to(gS.dtype).unsqueeze(-1), None


class ColorlessTerms(torch.autograd.Function):
    """(w_spec sum_g spectral_g + w_sparse sparsity(Q_last)) * inv_world and its two weighted parts
    (trainer.py:298-313) from S (G, K) and Q (G, n, n): spectral statistics kernel + one bookkeeping
    launch; both gradients are produced in the forward.  ``unit_grad``: the caller guarantees the
    upstream gradient of output[0] is 1 (loss.backward() on a plain sum), which skips two rescales."""

    @staticmethod
    def forward(ctx, S, Q, asym, w_spec, w_sparse, inv_world, unit_grad, nbands: int = 1):
        need = S.requires_grad or Q.requires_grad
        _, loss_g, gS = ops.spectral_stats(S, asym, w_spec * inv_world, want_grad=need)
        out, gQ = ops.colorless_terms(loss_g, Q, w_spec, w_sparse, inv_world, want_grad=need, nbands=nbands)
        ctx.set_materialize_grads(False)      # no zero-filled gradients for the two report outputs
        ctx.save_for_backward(gS, gQ)
        ctx.unit_grad = unit_grad
        if nbands > 1:
            if not unit_grad:
                raise ValueError("band-stacked colorless terms are back-propagated with unit gradients")
            # (nbands,) vectors: column views of the (nbands, 3) record, created inside the node
            total, spec, sparse = out[:, 0], out[:, 1], out[:, 2]
        else:
            # three 0-dim outputs of the node itself (indexing the result outside would put a
            # select-backward = zero fill + copy in front of the gradient)
            total, spec, sparse = out[0], out[1], out[2]
        ctx.mark_non_differentiable(spec, sparse)
        return total, spec, sparse

    @staticmethod
    def backward(ctx, g, _g1, _g2):
        gS, gQ = ctx.saved_tensors
        if g is None:
            return (None,) * 8
        if ctx.unit_grad:
            return gS, gQ, None, None, None, None, None, None
        return gS * g.to(gS.dtype), gQ * g, None, None, None, None, None, None


class SubFdnColorless(torch.autograd.Function):
    """Loss side of the colorless branch as ONE node (n <= 4; include/diffgfdn_hip.h "colorless side branch,
    fused"): group sums -> spectral losses + sparsity -> their gradients w.r.t. M (G, n, n) raw blocks, the
    gains b, c (N,) and the rotations Q (G, n, n).  ``Y, S, energy`` come from ``ops.subfdn_colorless_fwd``
    (the caller launches it first: the main branch hangs on its rescale of b, c -- Trainer.normalize,
    trainer.py:317-332, applied from the very responses the loss uses, one solve instead of two);
    ``normalized`` tells whether that call rescaled b, c.  Gradients are produced in the forward (the loss is
    a scalar head back-propagated with unit gradient); the backward only hands them over."""

    @staticmethod
    def forward(ctx, M, b, c, Q, Y, S, energy, grid: FrequencyGrid, delays, normalized, asym, w_spec, w_sparse,
                inv_world, nbands, want_grad):
        # (grad mode is always off inside Function.forward: the caller passes torch.is_grad_enabled())
        need = want_grad and (M.requires_grad or b.requires_grad or c.requires_grad)
        en = energy if normalized else None
        loss_g, gS = ops.spectral_stats_binmajor(S, en, asym, w_spec * inv_world, want_grad=need)
        out, gQ = ops.colorless_terms(loss_g, Q, w_spec, w_sparse, inv_world, want_grad=need, nbands=nbands)
        grads = (None, None, None)
        if need:
            grads = ops.subfdn_colorless_bwd(grid.turns, grid.logr, M, delays, b.data, c.data, en, Y, gS)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(gQ, *[g for g in grads if g is not None])
        ctx.shapes = (M.shape, b.shape, c.shape)
        if nbands > 1:
            total, spec, sparse = out[:, 0], out[:, 1], out[:, 2]
        else:
            total, spec, sparse = out[0], out[1], out[2]
        ctx.mark_non_differentiable(spec, sparse)
        return total, spec, sparse

    @staticmethod
    def backward(ctx, g, _g1, _g2):
        if g is None or len(ctx.saved_tensors) < 4:
            return (None,) * 16
        gQ, gM, gb, gc = ctx.saved_tensors         # unit upstream gradient by contract (see forward)
        sM, sb, sc = ctx.shapes
        return (gM.view(sM), gb.view(sb), gc.view(sc), gQ) + (None,) * 12


class IrfftOdd(torch.autograd.Function):
    """x = torch.fft.irfft(X, n) for odd n (losses.py:207-213, :442-445)."""

    @staticmethod
    def forward(ctx, X, n: int):
        ctx.n, ctx.ldx = n, X.shape[-1]
        return ops.irfft_odd_fwd(X, n)

    @staticmethod
    def backward(ctx, gx):
        return ops.irfft_odd_bwd(gx.contiguous(), ctx.n, ctx.ldx), None


class IrfftPow2(torch.autograd.Function):
    """x = torch.fft.irfft(X) with n = 2 (K-1) a power of two (utils.py:169, losses.py:344)."""

    @staticmethod
    def forward(ctx, X):
        n = 2 * (X.shape[-1] - 1)
        ctx.n = n
        return ops.irfft_pow2_fwd(X, n)

    @staticmethod
    def backward(ctx, gx):
        return ops.irfft_pow2_bwd(gx.contiguous(), ctx.n)


def irfft_like_torch(X: torch.Tensor, n: Optional[int] = None) -> torch.Tensor:
    """torch.fft.irfft(X, n) along the last axis for the two lengths the reference uses:
    n = X.shape[-1] odd, or the default n = 2 (K - 1) = 2^p."""
    shape = X.shape
    K = shape[-1]
    X2 = X.reshape(-1, K)
    if n is None:
        n = 2 * (K - 1)
    if n % 2 == 1:
        x = IrfftOdd.apply(X2, n)
    elif n == 2 * (K - 1) and (n & (n - 1)) == 0:
        x = IrfftPow2.apply(X2)
    else:
        raise NotImplementedError(f"irfft length {n} for {K} bins is not on the reference's path")
    return x.reshape(*shape[:-1], n)
