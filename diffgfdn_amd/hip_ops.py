"""Tensor-level wrappers of the C-ABI entry points (one python function per C function).

Every function takes / returns ``torch`` tensors that live on the GPU; outputs and scratch
buffers are allocated here with torch (PyTorch owns all memory, the library owns none) and the
launch goes onto ``torch.cuda.current_stream()``.  Inputs on the CPU raise: there is no CPU
fallback in the product path.
"""
import ctypes
from typing import Optional, Tuple

import torch

from . import _lib

_c64 = torch.complex64
_f32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("diffgfdn_amd HIP ops need CUDA/HIP tensors (no CPU fallback)")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(_f32).contiguous()


def _f64(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float64).contiguous()


def _c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(_c64).contiguous()


def _work(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------
def zprep(z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """z (K,) complex128 -> (turns, logr) float64."""
    _need_gpu(z)
    z = z.detach().to(torch.complex128).contiguous()
    K = z.numel()
    turns = torch.empty(K, dtype=torch.float64, device=z.device)
    logr = torch.empty(K, dtype=torch.float64, device=z.device)
    _lib.check(_lib.load().gfdn_zprep(_p(z), K, _p(turns), _p(logr), _stream()), "gfdn_zprep")
    return turns, logr


def ortho_fwd(M, want_Q=True, want_QQ=False):
    """M (G,n,n) -> (Q = expm(skew(M)), QQ = Q @ Q); entries not requested are None."""
    _need_gpu(M)
    M = _f(M)
    G, n, _ = M.shape
    Q = torch.empty_like(M) if want_Q else None
    QQ = torch.empty_like(M) if want_QQ else None
    _lib.check(_lib.load().gfdn_ortho_fwd(_p(M), G, n, _p(Q), _p(QQ), _stream()), "gfdn_ortho_fwd")
    return Q, QQ


def ortho_bwd(M, gQ=None, gQQ=None, Q=None):
    """``Q``: the forward's Q (G, n, n) if still at hand (skips one matrix exponential)."""
    _need_gpu(M)
    M = _f(M)
    G, n, _ = M.shape
    gQ = None if gQ is None else _f(gQ)
    gQQ = None if gQQ is None else _f(gQQ)
    if Q is not None:
        Q = _f(Q)
        if Q.shape != M.shape:
            raise RuntimeError("ortho_bwd: Q must have the shape of M")
    gM = torch.empty_like(M)
    _lib.check(_lib.load().gfdn_ortho_bwd(_p(M), G, n, _p(gQ), _p(gQQ), _p(Q), _p(gM), _stream()),
               "gfdn_ortho_bwd")
    return gM


def solve_fwd(turns, logr, A, delays, inv_gamma, b, transpose=False, inv_gamma_bins=None,
              precise: bool = False) -> torch.Tensor:
    """A (nblk,nper,nper) f32, delays/inv_gamma/b (N,) f32 -> Y (K,N) complex64.
    ``inv_gamma_bins`` (K, N) complex64 = 1 / Gamma_i(z_k): frequency-dependent absorption (``inv_gamma`` must
    then be ones).  ``precise``: matrix entries and elimination in float64 (ill-conditioned systems)."""
    _need_gpu(turns, A)
    A, delays, b = _f(A), _f(delays), _f(b)
    inv_gamma = _f64(inv_gamma) if precise else _f(inv_gamma)
    nblk, nper, _ = A.shape
    K = turns.numel()
    Y = torch.empty((K, nblk * nper), dtype=_c64, device=A.device)
    if inv_gamma_bins is not None:
        if precise:
            raise NotImplementedError("solve_fwd: precise with absorption filters")
        igz = _c(inv_gamma_bins)
        if tuple(igz.shape) != (K, nblk * nper):
            raise RuntimeError("solve_fwd: inv_gamma_bins must be (K, N)")
        _lib.check(_lib.load().gfdn_solve_absorb_fwd(_p(turns), _p(logr), K, nblk, nper, _p(A), _p(delays),
                                                     _p(inv_gamma), _p(igz), _p(b), int(transpose), _p(Y),
                                                     _stream()), "gfdn_solve_absorb_fwd")
        return Y
    lib = _lib.load()
    fn, name = (lib.gfdn_solve_precise_fwd, "gfdn_solve_precise_fwd") if precise else (lib.gfdn_solve_fwd, "gfdn_solve_fwd")
    _lib.check(fn(_p(turns), _p(logr), K, nblk, nper, _p(A), _p(delays), _p(inv_gamma), _p(b), int(transpose), _p(Y),
                  _stream()), name)
    return Y


def solve_bwd(turns, logr, A, delays, inv_gamma, b, gY, transpose=False, Y=None, inv_gamma_bins=None,
              precise: bool = False):
    """-> gA (nblk,nper,nper), gb (N,), ginv_gamma (N,)  (float32).  ``Y``: the forward solution
    (K, N) if still at hand -- the kernel then does not re-solve the forward system."""
    _need_gpu(turns, A, gY)
    A, delays, b, gY = _f(A), _f(delays), _f(b), _c(gY)
    inv_gamma = _f64(inv_gamma) if precise else _f(inv_gamma)
    nblk, nper, _ = A.shape
    K = turns.numel()
    if Y is not None:
        Y = _c(Y)
        if tuple(Y.shape) != (K, nblk * nper):
            raise RuntimeError("solve_bwd: Y must be the (K, N) forward solution")
    lib = _lib.load()
    gA = torch.empty_like(A)
    gb = torch.empty(nblk * nper, dtype=_f32, device=A.device)
    gig = torch.empty_like(gb)
    work = _work(lib.gfdn_solve_bwd_work_bytes(nblk, nper), A.device)
    if inv_gamma_bins is not None:
        igz = _c(inv_gamma_bins)
        if tuple(igz.shape) != (K, nblk * nper):
            raise RuntimeError("solve_bwd: inv_gamma_bins must be (K, N)")
        _lib.check(lib.gfdn_solve_absorb_bwd(_p(turns), _p(logr), K, nblk, nper, _p(A), _p(delays), _p(inv_gamma),
                                             _p(igz), _p(b), int(transpose), _p(gY), _p(Y), _p(gA), _p(gb), _p(gig),
                                             _p(work), _stream()), "gfdn_solve_absorb_bwd")
        return gA, gb, None
    fn, name = (lib.gfdn_solve_precise_bwd, "gfdn_solve_precise_bwd") if precise else (lib.gfdn_solve_bwd, "gfdn_solve_bwd")
    _lib.check(fn(_p(turns), _p(logr), K, nblk, nper, _p(A), _p(delays), _p(inv_gamma), _p(b), int(transpose),
                  _p(gY), _p(Y), _p(gA), _p(gb), _p(gig), _p(work), _stream()), name)
    return gA, gb, gig


def solve_phi_fwd(turns, logr, BM, Phi, nper, delays, inv_gamma, b, inv_gamma_bins=None) -> torch.Tensor:
    """FILTER coupling: A(z_k) = BM o kron(Phi_k, 1).  BM (N, N) f32, Phi (K, G, G) complex64 -> Y (K, N).
    ``inv_gamma_bins`` (K, N) complex64: 1 / Gamma_i(z_k) of absorption filters on the lines (multiplies inv_gamma)."""
    _need_gpu(turns, BM, Phi)
    BM, delays, inv_gamma, b, Phi = _f(BM), _f(delays), _f(inv_gamma), _f(b), _c(Phi)
    K, G = turns.numel(), Phi.shape[-1]
    if tuple(Phi.shape) != (K, G, G) or tuple(BM.shape) != (G * nper, G * nper):
        raise RuntimeError("solve_phi_fwd: Phi must be (K, G, G) and BM (G nper, G nper)")
    igb = None
    if inv_gamma_bins is not None:
        igb = _c(inv_gamma_bins)
        if tuple(igb.shape) != (K, G * nper):
            raise RuntimeError("solve_phi_fwd: inv_gamma_bins must be (K, N)")
    Y = torch.empty((K, G * nper), dtype=_c64, device=BM.device)
    _lib.check(_lib.load().gfdn_solve_phi_absorb_fwd(_p(turns), _p(logr), K, G, nper, _p(BM), _p(Phi), _p(delays),
                                                     _p(inv_gamma), _p(igb), _p(b), _p(Y), _stream()),
               "gfdn_solve_phi_absorb_fwd")
    return Y


def solve_phi_bwd(turns, logr, BM, Phi, nper, delays, inv_gamma, b, gY, Y, inv_gamma_bins=None):
    """-> gBM (N, N), gb (N,), ginv_gamma (N,) float32 and gPhi (K, G, G) complex64 (per bin)."""
    _need_gpu(turns, BM, Phi, gY, Y)
    BM, delays, inv_gamma, b, Phi, gY, Y = _f(BM), _f(delays), _f(inv_gamma), _f(b), _c(Phi), _c(gY), _c(Y)
    K, G = turns.numel(), Phi.shape[-1]
    N = G * nper
    igb = None if inv_gamma_bins is None else _c(inv_gamma_bins)
    if tuple(Phi.shape) != (K, G, G) or tuple(BM.shape) != (N, N) or tuple(Y.shape) != (K, N) \
            or tuple(gY.shape) != (K, N) or (igb is not None and tuple(igb.shape) != (K, N)):
        raise RuntimeError("solve_phi_bwd: shape mismatch")
    lib = _lib.load()
    gBM = torch.empty_like(BM)
    gb = torch.empty(N, dtype=_f32, device=BM.device)
    gig = torch.empty_like(gb)
    gPhi = torch.empty_like(Phi)
    work = _work(lib.gfdn_solve_phi_bwd_work_bytes(G, nper), BM.device)
    _lib.check(lib.gfdn_solve_phi_absorb_bwd(_p(turns), _p(logr), K, G, nper, _p(BM), _p(Phi), _p(delays),
                                             _p(inv_gamma), _p(igb), _p(b), _p(gY), _p(Y), _p(gBM), _p(gb), _p(gig),
                                             _p(gPhi), _p(work), _stream()), "gfdn_solve_phi_absorb_bwd")
    return gBM, gb, gig, gPhi


def _z128(z: torch.Tensor) -> torch.Tensor:
    return z.detach().to(torch.complex128).contiguous()


def svf_coefficients(raw, cutoff, compress_pole_factor: float, gcoef=None) -> torch.Tensor:
    """raw (..., S, 2) f32 unconstrained SVF parameters, cutoff (S,) float64 -> biquad coefficients (..., S, 6) f32;
    with ``gcoef`` (..., S, 6): the gradient with respect to raw instead."""
    _need_gpu(raw, cutoff)
    raw = _f(raw)
    cutoff = _f64(cutoff)
    S = raw.shape[-2]
    R = raw.numel() // (2 * S)
    if raw.shape[-1] != 2 or cutoff.numel() != S:
        raise RuntimeError("svf_coefficients: raw (..., S, 2), cutoff (S,)")
    if gcoef is not None:
        gcoef = _f(gcoef)
        if tuple(gcoef.shape) != tuple(raw.shape[:-1]) + (6,):
            raise RuntimeError("svf_coefficients: gcoef must be (..., S, 6)")
        out = torch.empty_like(raw)
    else:
        out = torch.empty(tuple(raw.shape[:-1]) + (6,), dtype=_f32, device=raw.device)
    _lib.check(_lib.load().gfdn_svf_coefficients(_p(raw), _p(cutoff), float(compress_pole_factor), R, S, _p(gcoef),
                                                 _p(out), _stream()), "gfdn_svf_coefficients")
    return out


def sos_response(coef, z) -> torch.Tensor:
    """coef (R, S, 6) f32 [b0 b1 b2 a0 a1 a2], z (K,) complex -> (R, K) complex64 responses of the R cascades
    (sections in float64, rounded to complex64, running product in complex64)."""
    _need_gpu(coef, z)
    coef, z = _f(coef), _z128(z)
    R, S, six = coef.shape
    if six != 6:
        raise RuntimeError("sos_response: coef must be (R, S, 6)")
    out = torch.empty((R, z.numel()), dtype=_c64, device=coef.device)
    _lib.check(_lib.load().gfdn_sos_response(_p(coef), R, S, _p(z), z.numel(), _p(out), _stream()), "gfdn_sos_response")
    return out


def sos_compose_fwd(coef, z, T, direct=None) -> torch.Tensor:
    """H[b][k] = sum_g cascade_{b,g}(z_k) T[k][g] + direct[b][k]: coef (B, G, S, 6) f32, T (K, G) c64,
    direct (B, K) c64 or None -> (B, K) c64."""
    _need_gpu(coef, z, T)
    coef, z, T = _f(coef), _z128(z), _c(T)
    B, G, S, six = coef.shape
    K = z.numel()
    if six != 6 or tuple(T.shape) != (K, G):
        raise RuntimeError("sos_compose_fwd: coef (B, G, S, 6), T (K, G)")
    if direct is not None:
        direct = _c(direct)
        if tuple(direct.shape) != (B, K):
            raise RuntimeError("sos_compose_fwd: direct must be (B, K)")
    H = torch.empty((B, K), dtype=_c64, device=coef.device)
    _lib.check(_lib.load().gfdn_sos_compose_fwd(_p(coef), B, G, S, _p(z), K, _p(T), _p(direct), K, _p(H), _stream()),
               "gfdn_sos_compose_fwd")
    return H


def sos_compose_bwd(coef, z, T, gH):
    """-> gcoef (B, G, S, 6) f32, gT (K, G) c64."""
    _need_gpu(coef, z, T, gH)
    coef, z, T, gH = _f(coef), _z128(z), _c(T), _c(gH)
    B, G, S, _ = coef.shape
    K = z.numel()
    lib = _lib.load()
    tc, cc = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.gfdn_sos_compose_bwd_chunks(B, K, ctypes.byref(tc), ctypes.byref(cc)), "gfdn_sos_compose_bwd_chunks")
    gT_part = torch.empty((tc.value, K, G), dtype=_c64, device=coef.device)
    gc_part = torch.empty((B * G, cc.value, S, 6), dtype=_f32, device=coef.device)
    _lib.check(lib.gfdn_sos_compose_bwd(_p(coef), B, G, S, _p(z), K, _p(T), _p(gH), _p(gT_part), _p(gc_part),
                                        _stream()), "gfdn_sos_compose_bwd")
    return gc_part.sum(1).reshape(B, G, S, 6), gT_part.sum(0)


def _rows(rows, n_items: int, store_rows: int):
    """Validate a row-indirection index (int64 device vector of n_items entries)."""
    if rows is None:
        return None
    _need_gpu(rows)
    if rows.dtype != torch.int64 or rows.dim() != 1 or rows.numel() != n_items or not rows.is_contiguous():
        raise RuntimeError("row indirection: expected a contiguous int64 device vector with one entry per item")
    if store_rows <= 0:
        raise RuntimeError("row indirection: empty store")
    return rows


def compose_fwd(Y, c, rgain, nper, direct=None, filt=None, want_S=False, direct_rows=None,
                nbands: int = 1):
    """Y (K,N) c64, c (N,), rgain (B,G) -> H (B,K) c64 [, S (G,K) c64].
    ``direct_rows``: item b adds row direct_rows[b] of ``direct`` (a store of all receivers).
    ``nbands`` > 1: band-stacked (include/diffgfdn_hip.h): Y (K, nbands*N), c (nbands*N,), rgain
    (nbands*Bper, G) band-major items, filt (nbands, K) -> H (nbands*Bper, K) [, S (nbands*G, K)]."""
    _need_gpu(Y, c, rgain)
    Y, c, rgain = _c(Y), _f(c), _f(rgain)
    K, Ntot = Y.shape
    Btot, G = rgain.shape
    if Ntot != nbands * G * nper or Btot % nbands or c.numel() != Ntot:
        raise RuntimeError("compose_fwd: shapes do not match nbands x (G x nper) delay lines")
    B = Btot // nbands
    ldd = 0
    if direct is not None:
        direct = direct.detach()
        # a row-strided view (e.g. the first K bins of longer rows) is consumed in place
        if not (direct.dtype == _c64 and direct.dim() == 2 and direct.stride(1) == 1
                and direct.stride(0) >= K and direct.shape[1] >= K):
            direct = _c(direct)
        ldd = direct.stride(0)
    filt = None if filt is None else _c(filt)
    if filt is not None and filt.numel() != nbands * K:
        raise RuntimeError("compose_fwd: filt must hold K bins per band")
    direct_rows = None if direct is None else _rows(direct_rows, Btot, direct.shape[0])
    if direct is not None and direct_rows is None and direct.shape[0] != Btot:
        raise RuntimeError("compose_fwd: direct must have one row per item (or pass direct_rows)")
    H = torch.empty((Btot, K), dtype=_c64, device=Y.device)
    S = torch.empty((nbands * G, K), dtype=_c64, device=Y.device) if want_S else None
    if nbands == 1:
        _lib.check(_lib.load().gfdn_compose_fwd(_p(Y), K, G, nper, _p(c), _p(rgain), B, _p(direct),
                                                ldd, _p(direct_rows), _p(filt), _p(H), K,
                                                _p(S), _stream()), "gfdn_compose_fwd")
    else:
        _lib.check(_lib.load().gfdn_compose_banded_fwd(_p(Y), K, nbands, G, nper, _p(c), _p(rgain), B,
                                                       _p(direct), ldd, _p(direct_rows), _p(filt), K,
                                                       _p(H), K, _p(S), _stream()),
                   "gfdn_compose_banded_fwd")
    return (H, S) if want_S else H


def compose_bwd(Y, c, rgain, nper, gH, filt=None, nbands: int = 1):
    """-> gY (K,N) c64, gc (N,), grgain (B,G)  (band-stacked shapes as in compose_fwd)."""
    _need_gpu(Y, gH)
    Y, c, rgain, gH = _c(Y), _f(c), _f(rgain), _c(gH)
    K, N = Y.shape
    Btot, G = rgain.shape
    if N != nbands * G * nper or Btot % nbands or tuple(gH.shape) != (Btot, K):
        raise RuntimeError("compose_bwd: shapes do not match nbands x (G x nper) delay lines")
    B = Btot // nbands
    filt = None if filt is None else _c(filt)
    lib = _lib.load()
    gY = torch.empty_like(Y)
    gc = torch.empty(N, dtype=_f32, device=Y.device)
    grg = torch.empty((Btot, G), dtype=_f32, device=Y.device)
    if nbands == 1:
        work = _work(lib.gfdn_compose_bwd_work_bytes(K, G, nper, B), Y.device)
        _lib.check(lib.gfdn_compose_bwd(_p(Y), K, G, nper, _p(c), _p(rgain), B, _p(filt), _p(gH), K,
                                        _p(gY), _p(gc), _p(grg), _p(work), _stream()),
                   "gfdn_compose_bwd")
    else:
        work = _work(lib.gfdn_compose_banded_bwd_work_bytes(K, nbands, G, nper, B), Y.device)
        _lib.check(lib.gfdn_compose_banded_bwd(_p(Y), K, nbands, G, nper, _p(c), _p(rgain), B, _p(filt), K,
                                               _p(gH), K, _p(gY), _p(gc), _p(grg), _p(work), _stream()),
                   "gfdn_compose_banded_bwd")
    return gY, gc, grg


def compose_sh_fwd(Y, c, w, G, nper, filt=None):
    """Y (K,N), c (N,), w (B,G,nper) -> H_sh (B,nper,K) c64."""
    _need_gpu(Y, w)
    Y, c, w = _c(Y), _f(c), _f(w)
    K, N = Y.shape
    B = w.shape[0]
    filt = None if filt is None else _c(filt)
    H = torch.empty((B, nper, K), dtype=_c64, device=Y.device)
    _lib.check(_lib.load().gfdn_compose_sh_fwd(_p(Y), K, G, nper, _p(c), _p(w), B, _p(filt), _p(H),
                                               _stream()), "gfdn_compose_sh_fwd")
    return H


def compose_sh_bwd(Y, c, w, G, nper, gH, filt=None):
    """-> gY (K,N), gc (N,), gw (B,G,nper)."""
    _need_gpu(Y, gH)
    Y, c, w, gH = _c(Y), _f(c), _f(w), _c(gH)
    K, N = Y.shape
    B = w.shape[0]
    filt = None if filt is None else _c(filt)
    lib = _lib.load()
    gY = torch.empty_like(Y)
    gc = torch.empty(N, dtype=_f32, device=Y.device)
    gw = torch.empty((B, G, nper), dtype=_f32, device=Y.device)
    work = _work(lib.gfdn_compose_sh_bwd_work_bytes(G, nper, B), Y.device)
    _lib.check(lib.gfdn_compose_sh_bwd(_p(Y), K, G, nper, _p(c), _p(w), B, _p(filt), _p(gH), _p(gY),
                                       _p(gc), _p(gw), _p(work), _stream()), "gfdn_compose_sh_bwd")
    return gY, gc, gw


def spectral_stats(S, asym: bool, scale: float = 1.0, want_grad: bool = True):
    """S (G,K) c64 -> energy (G,), loss (G,), gS (G,K) or None."""
    _need_gpu(S)
    S = _c(S)
    G, K = S.shape
    energy = torch.empty(G, dtype=_f32, device=S.device)
    loss = torch.empty(G, dtype=_f32, device=S.device)
    gS = torch.empty_like(S) if want_grad else None
    lib = _lib.load()
    work = _work(lib.gfdn_spectral_stats_work_bytes(G, K), S.device)
    _lib.check(lib.gfdn_spectral_stats(_p(S), G, K, int(asym), float(scale), _p(energy), _p(loss),
                                       _p(gS), _p(work), _stream()), "gfdn_spectral_stats")
    return energy, loss, gS


def colorless_terms(loss_g, Q, w_spec, w_sparse, inv_world, want_grad=True, nbands: int = 1):
    """-> out3 = [(spectral + sparsity) * inv_world, spectral, sparsity], gQ (G,n,n) or None.
    ``nbands`` > 1: loss_g (nbands*G,), Q (nbands*G,n,n) -> out (nbands, 3)."""
    _need_gpu(loss_g, Q)
    loss_g, Q = _f(loss_g), _f(Q)
    Gtot, n, _ = Q.shape
    if Gtot % nbands or loss_g.numel() != Gtot:
        raise RuntimeError("colorless_terms: one spectral loss per group, groups divisible by nbands")
    G = Gtot // nbands
    gQ = torch.empty_like(Q) if want_grad else None
    if nbands == 1:
        out = torch.empty(3, dtype=_f32, device=Q.device)
        _lib.check(_lib.load().gfdn_colorless_terms(_p(loss_g), G, _p(Q), n, float(w_spec), float(w_sparse),
                                                    float(inv_world), _p(out), _p(gQ), _stream()),
                   "gfdn_colorless_terms")
    else:
        out = torch.empty((nbands, 3), dtype=_f32, device=Q.device)
        _lib.check(_lib.load().gfdn_colorless_terms_banded(_p(loss_g), nbands, G, _p(Q), n, float(w_spec),
                                                           float(w_sparse), float(inv_world), _p(out),
                                                           _p(gQ), _stream()), "gfdn_colorless_terms_banded")
    return out, gQ


def weighted_sums(a, wa, b, wb, a_div=None, a_rows=None, nbands: int = 1):
    """-> [wa sum(a) + wb sum(b), wa sum(a), wb sum(b)] (float32, 3).  ``a`` may be (n, cols) partial
    sums per item (edr_loss(defer=True)), divided per item by a_div[a_rows[i]].
    ``nbands`` > 1: the n items are band-major, -> (nbands, 3) sums per band."""
    ref = a if a is not None else b
    _need_gpu(ref)
    n = ref.shape[0] if ref.dim() == 2 else ref.numel()
    cols = a.shape[1] if (a is not None and a.dim() == 2) else 1
    if a is not None:
        a = _f(a)
    if a is not None and b is not None and b.numel() != n:
        raise RuntimeError("weighted_sums: a and b must hold the same number of items")
    a_rows = None if a_div is None else _rows(a_rows, n, a_div.numel())
    if a_div is not None and a_rows is None and a_div.numel() != n:
        raise RuntimeError("weighted_sums: a_div must have one entry per item (or pass a_rows)")
    if nbands > 1:
        if n % nbands:
            raise RuntimeError("weighted_sums: items must divide evenly into bands")
        out = torch.empty((nbands, 3), dtype=_f32, device=ref.device)
        _lib.check(_lib.load().gfdn_weighted_sums_banded(_p(a), cols, _p(a_div), _p(a_rows), float(wa), _p(b),
                                                         float(wb), n // nbands, nbands, _p(out), _stream()),
                   "gfdn_weighted_sums_banded")
        return out
    out = torch.empty(3, dtype=_f32, device=ref.device)
    _lib.check(_lib.load().gfdn_weighted_sums(_p(a), cols, _p(a_div), _p(a_rows), float(wa), _p(b),
                                              float(wb), n, _p(out), _stream()), "gfdn_weighted_sums")
    return out


def subfdn_normalize(turns, logr, M, delays, b, c, want_energy: bool = False):
    """Trainer.normalize in two launches: in place b[n], c[n] /= E_g^(1/4) with E_g the mean energy of
    group g's sub-FDN response (raw blocks M (G,nper,nper), no absorption).  Returns E (G,) or None."""
    _need_gpu(turns, M, b, c)
    M, delays = _f(M), _f(delays)
    G, nper, _ = M.shape
    for t in (b, c):
        if t.dtype != _f32 or not t.is_contiguous() or t.numel() != G * nper:
            raise RuntimeError("subfdn_normalize: gains must be contiguous float32 of G*nper elements")
    lib = _lib.load()
    energy = torch.empty(G, dtype=_f32, device=M.device) if want_energy else None
    work = _work(lib.gfdn_subfdn_normalize_work_bytes(G), M.device)
    _lib.check(lib.gfdn_subfdn_normalize(_p(turns), _p(logr), turns.numel(), G, nper, _p(M), _p(delays),
                                         _p(b), _p(c), _p(energy), _p(work), _stream()),
               "gfdn_subfdn_normalize")
    return energy


def subfdn_colorless_fwd(turns, logr, M, delays, b, c, normalize: bool):
    """Fused sub-FDN forward of the colorless branch (nper <= 4): raw responses Y (K, G*nper), raw group sums
    S (K, G) BIN-MAJOR, energy (G) = mean_k |S|^2; ``normalize``: b, c /= energy^(1/4) IN PLACE."""
    _need_gpu(turns, M, b, c)
    M, delays = _f(M), _f(delays)
    G, nper, _ = M.shape
    for t in (b, c):
        if t.dtype != _f32 or not t.is_contiguous() or t.numel() != G * nper:
            raise RuntimeError("subfdn_colorless_fwd: gains must be contiguous float32 of G*nper elements")
    K = turns.numel()
    lib = _lib.load()
    Y = torch.empty((K, G * nper), dtype=_c64, device=M.device)
    S = torch.empty((K, G), dtype=_c64, device=M.device)
    energy = torch.empty(G, dtype=_f32, device=M.device)
    work = _work(lib.gfdn_subfdn_colorless_work_bytes(G, nper), M.device)
    _lib.check(lib.gfdn_subfdn_colorless_fwd(_p(turns), _p(logr), K, G, nper, _p(M), _p(delays), _p(b), _p(c),
                                             int(bool(normalize)), _p(Y), _p(S), _p(energy), _p(work), _stream()),
               "gfdn_subfdn_colorless_fwd")
    return Y, S, energy


def spectral_stats_binmajor(S, energy, asym: bool, scale: float = 1.0, want_grad: bool = True):
    """S (K, G) bin-major raw group sums, energy (G) or None -> loss (G,) on S' = S / sqrt(energy), gS (K, G)."""
    _need_gpu(S)
    S = _c(S)
    K, G = S.shape
    loss = torch.empty(G, dtype=_f32, device=S.device)
    gS = torch.empty_like(S) if want_grad else None
    lib = _lib.load()
    work = _work(lib.gfdn_subfdn_colorless_work_bytes(G, 4), S.device)
    _lib.check(lib.gfdn_spectral_stats_binmajor(_p(S), G, K, _p(energy), int(asym), float(scale), _p(loss), _p(gS),
                                                _p(work), _stream()), "gfdn_spectral_stats_binmajor")
    return loss, gS


def subfdn_colorless_bwd(turns, logr, M, delays, b, c, energy, Y, gS):
    """-> gM (G,nper,nper), gb (N,), gc (N,) of sum_g <gS[:, g], S'[:, g]> at the current (rescaled) b, c."""
    _need_gpu(turns, M, Y, gS)
    M, delays, b, c = _f(M), _f(delays), _f(b), _f(c)
    G, nper, _ = M.shape
    K = turns.numel()
    if tuple(Y.shape) != (K, G * nper) or tuple(gS.shape) != (K, G):
        raise RuntimeError("subfdn_colorless_bwd: Y (K, G*nper) and gS (K, G) expected")
    lib = _lib.load()
    gM = torch.empty_like(M)
    gb = torch.empty(G * nper, dtype=_f32, device=M.device)
    gc = torch.empty_like(gb)
    work = _work(lib.gfdn_subfdn_colorless_work_bytes(G, nper), M.device)
    _lib.check(lib.gfdn_subfdn_colorless_bwd(_p(turns), _p(logr), K, G, nper, _p(M), _p(delays), _p(b), _p(c),
                                             _p(energy), _p(_c(Y)), _p(_c(gS)), _p(gM), _p(gb), _p(gc), _p(work),
                                             _stream()), "gfdn_subfdn_colorless_bwd")
    return gM, gb, gc


def normalize_io(energy, b, c, G: int, nper: int):
    """In place: b[n], c[n] /= energy[group(n)]^(1/4)  (b, c float32 contiguous, N = G*nper)."""
    _need_gpu(energy, b, c)
    for t in (b, c):
        if t.dtype != _f32 or not t.is_contiguous() or t.numel() != G * nper:
            raise RuntimeError("normalize_io: gains must be contiguous float32 of G*nper elements")
    _lib.check(_lib.load().gfdn_normalize_io(_p(energy), _p(b), _p(c), G, nper, _stream()),
               "gfdn_normalize_io")


# ------------------------------------------------------------------------------------------------
# block transfer functions in polynomial form (csrc/blocktf.hip; nper <= 4, zero coupling)
def tf_coefs(A, b, c, inv_gamma=None, out=None) -> torch.Tensor:
    """A (nblk, n, n), b, c (nblk*n), inv_gamma (nblk*n) or None -> coef (nblk, 32) float32 records
    [P_S | Q_S] with T(z) = sum_S P_S e_S / sum_S Q_S e_S, e_S = prod_{i in S} z^{m_i}."""
    _need_gpu(A, b, c)
    A, b, c = _f(A), _f(b).reshape(-1), _f(c).reshape(-1)
    nblk, n, _ = A.shape
    if n > 4:
        raise RuntimeError("tf_coefs: blocks of at most 4 delay lines")
    if b.numel() != nblk * n or c.numel() != nblk * n:
        raise RuntimeError("tf_coefs: b, c must hold nblk * n gains")
    ig = None if inv_gamma is None else _f(inv_gamma).reshape(-1)
    if ig is not None and ig.numel() != nblk * n:
        raise RuntimeError("tf_coefs: inv_gamma must hold nblk * n gains")
    coef = torch.empty((nblk, 32), dtype=_f32, device=A.device) if out is None else out
    _lib.check(_lib.load().gfdn_tf_coefs_fwd(_p(A), _p(b), _p(c), _p(ig), nblk, n, _p(coef), _stream()),
               "gfdn_tf_coefs_fwd")
    return coef


def tf_coefs2(A0, ig0, A1, ig1, b, c):
    """Records of two sets of blocks sharing b, c in one launch -> (coef0, coef1), each (nblk, 32)."""
    _need_gpu(A0, A1, b, c)
    A0, A1, b, c = _f(A0), _f(A1), _f(b).reshape(-1), _f(c).reshape(-1)
    nblk, n, _ = A0.shape
    if n > 4 or A1.shape != A0.shape or b.numel() != nblk * n or c.numel() != nblk * n:
        raise RuntimeError("tf_coefs2: two sets of nblk blocks of at most 4 lines, b, c of nblk * n gains")
    ig0 = None if ig0 is None else _f(ig0).reshape(-1)
    ig1 = None if ig1 is None else _f(ig1).reshape(-1)
    c0 = torch.empty((nblk, 32), dtype=_f32, device=A0.device)
    c1 = torch.empty((nblk, 32), dtype=_f32, device=A0.device)
    _lib.check(_lib.load().gfdn_tf_coefs_fwd2(_p(A0), _p(ig0), _p(c0), _p(A1), _p(ig1), _p(c1), _p(b), _p(c), nblk, n,
                                              _stream()), "gfdn_tf_coefs_fwd2")
    return c0, c1


def tf_parts(K: int, nblk: int) -> int:
    n = _lib.load().gfdn_tf_parts(int(K), int(nblk))
    if n <= 0:
        raise RuntimeError("tf_parts: unsupported number of blocks")
    return n


def tf_gpart(nblk: int, device) -> torch.Tensor:
    """Partial-record buffer (max parts, nblk, 32) float32 of the reducing launches."""
    return torch.empty(_lib.load().gfdn_tf_gpart_bytes(nblk) // 4, dtype=_f32, device=device)


def tf_eval(turns, logr, coef, delays, nper: int, scale=None) -> torch.Tensor:
    """-> T (K, nblk) complex64 bin-major."""
    _need_gpu(turns, coef, delays)
    coef, delays = _f(coef), _f(delays)
    nblk, K = coef.shape[0], turns.numel()
    if delays.numel() != nblk * nper:
        raise RuntimeError("tf_eval: delays must hold nblk * nper entries")
    scale = None if scale is None else _f(scale)
    T = torch.empty((K, nblk), dtype=_c64, device=coef.device)
    _lib.check(_lib.load().gfdn_tf_eval(_p(turns), _p(logr), K, nblk, nper, _p(coef), _p(delays), _p(scale), _p(T),
                                        _stream()), "gfdn_tf_eval")
    return T


def tf_energy(turns, logr, coef, delays, nper: int, b=None, c=None, want_energy=True, want_scale=True, work=None,
              phase: int = 3, energy=None, scale=None, dturn: float = 0.0, gains=None, G: int = 0):
    """energy (nblk) = mean_k |T|^2, scale = energy^(-1/2); b, c (float32, contiguous): rescaled in place.
    ``phase``: 1 = only the pass over the bins (partial sums into ``work``; returns work), 2 = only the finish
    from that ``work``, 3 = both.  ``dturn`` != 0: the grid is uniform on the unit circle with that step (turns).
    ``gains`` (bands * Bper, G) with nblk = bands * G: returns (energy, scale, gains_scaled) with gains_scaled[band Bper + r][g]
    = gains[...][g] scale[band G + g], written by the finish launch (the scale folded into the receiver gains)."""
    _need_gpu(turns, coef, delays)
    coef, delays = _f(coef), _f(delays)
    nblk, K = coef.shape[0], turns.numel()
    for t in (b, c):
        if t is not None and (t.dtype != _f32 or not t.is_contiguous() or t.numel() != nblk * nper):
            raise RuntimeError("tf_energy: gains must be contiguous float32 of nblk*nper elements")
    lib = _lib.load()
    if phase & 2:
        energy = torch.empty(nblk, dtype=_f32, device=coef.device) if (want_energy and energy is None) else energy
        scale = torch.empty(nblk, dtype=_f32, device=coef.device) if (want_scale and scale is None) else scale
    work = _work(lib.gfdn_tf_work_bytes(nblk), coef.device) if work is None else work
    if gains is not None:
        if not (phase & 2) or G <= 0 or nblk % G or gains.dtype != _f32 or not gains.is_contiguous() or gains.shape[1] != G \
                or gains.shape[0] % (nblk // G):
            raise RuntimeError("tf_energy: gains (bands * Bper, G) float32 contiguous with nblk = bands * G, with the finish")
        gs = torch.empty_like(gains)
        _lib.check(lib.gfdn_tf_energy_gains(_p(turns), _p(logr), K, nblk, nper, _p(coef), _p(delays), _p(b), _p(c), _p(energy),
                                            _p(scale), _p(work), int(phase), float(dturn), _p(gains), _p(gs),
                                            gains.shape[0] // (nblk // G), G, _stream()), "gfdn_tf_energy_gains")
        return energy, scale, gs
    _lib.check(lib.gfdn_tf_energy(_p(turns), _p(logr), K, nblk, nper, _p(coef), _p(delays), _p(b), _p(c), _p(energy),
                                  _p(scale), _p(work), int(phase), float(dturn), _stream()), "gfdn_tf_energy")
    return work if phase == 1 else (energy, scale)


def tf_colorless(turns, logr, coef, delays, nper: int, scale, asym: bool, gscale: float, work=None,
                 dturn: float = 0.0):
    """-> (grec (nblk, 32) gradient records of gscale * sum_blk loss_blk for the (scaled) sub-FDN records,
    loss (nblk,) = mean_k (|scale T| - 1)^p)."""
    _need_gpu(turns, coef, delays)
    coef, delays = _f(coef), _f(delays)
    nblk, K = coef.shape[0], turns.numel()
    scale = None if scale is None else _f(scale)
    work = tf_gpart(nblk, coef.device) if work is None else work
    grec = torch.empty((nblk, 32), dtype=_f32, device=coef.device)
    loss = torch.empty(nblk, dtype=_f32, device=coef.device)
    _lib.check(_lib.load().gfdn_tf_colorless(_p(turns), _p(logr), K, nblk, nper, _p(coef), _p(delays), _p(scale),
                                             int(asym), float(gscale), _p(grec), _p(loss), _p(work), float(dturn),
                                             _stream()),
               "gfdn_tf_colorless")
    return grec, loss


def tf_compose_fwd(turns, logr, coef, delays, nper: int, rgain, scale=None, direct=None, filt=None,
                   direct_rows=None, nbands: int = 1, out=None, save_T: bool = False, want_H: bool = True):
    """H (nbands*B, K) complex64 from the records (band-stacked as compose_fwd); ``save_T``: also returns the
    scaled, unfiltered group transfer functions (nbands*G, K) complex64 that tf_compose_bwd takes back.
    ``want_H=False`` (with save_T): the transfer functions only -- returns (Tquad (nbands, K, 4), Tsave), Tquad being
    what irfft_odd_pairs_compose_fwd takes (it forms H inside the transform)."""
    _need_gpu(turns, coef, rgain)
    coef, delays, rgain = _f(coef), _f(delays), _f(rgain)
    K = turns.numel()
    Btot, G = rgain.shape
    if coef.shape[0] != nbands * G or Btot % nbands or delays.numel() != nbands * G * nper:
        raise RuntimeError("tf_compose_fwd: shapes do not match nbands x G blocks")
    ldd = 0
    if direct is not None:
        direct = direct.detach()
        if not (direct.dtype == _c64 and direct.dim() == 2 and direct.stride(1) == 1 and direct.stride(0) >= K
                and direct.shape[1] >= K):
            direct = _c(direct)
        ldd = direct.stride(0)
    filt = None if filt is None else _c(filt)
    if filt is not None and filt.numel() != nbands * K:
        raise RuntimeError("tf_compose_fwd: filt must hold K bins per band")
    direct_rows = None if direct is None else _rows(direct_rows, Btot, direct.shape[0])
    if direct is not None and direct_rows is None and direct.shape[0] != Btot:
        raise RuntimeError("tf_compose_fwd: direct must have one row per item (or pass direct_rows)")
    scale = None if scale is None else _f(scale)
    if not want_H and not save_T:
        raise RuntimeError("tf_compose_fwd: nothing to compute")
    H = None if not want_H else (torch.empty((Btot, K), dtype=_c64, device=coef.device) if out is None else out)
    Ts = torch.empty((nbands * G, K), dtype=_c64, device=coef.device) if save_T else None
    # without H: also the band's transfer functions of a bin side by side (nbands, K, 4), for the transform's first pass
    Tq = torch.empty((nbands, K, 4), dtype=_c64, device=coef.device) if not want_H else None
    _lib.check(_lib.load().gfdn_tf_compose_fwd(_p(turns), _p(logr), K, nbands, G, nper, _p(coef), _p(delays),
                                               _p(scale), _p(rgain), Btot // nbands, _p(direct), ldd,
                                               _p(direct_rows), _p(filt), K, _p(H), K, _p(Ts), _p(Tq), _stream()),
               "gfdn_tf_compose_fwd")
    if not want_H:
        return Tq, Ts
    return (H, Ts) if save_T else H


def tf_gain_grad(Tsave, gH, G: int, filt=None, nbands: int = 1, grgain=None, work=None, partial: bool = False):
    """grgain (nbands*B, G) = sum_k Re(dL/dH[b][k] conj(filt[k] T'_g[k])); ``Tsave`` (nbands*G, K) from
    tf_compose_fwd(save_T=True).  ``partial``: skip the sums over the bin chunks and return the (nbands*B*G, chunks)
    partial rows (a buffer of their own) for ``mlp_gains_bwd(ggains_parts=...)``, which sums them itself."""
    _need_gpu(Tsave, gH)
    Tsave, gH = _c(Tsave), _c(gH)
    Btot, K = gH.shape
    if Btot % nbands or tuple(Tsave.shape) != (nbands * G, K):
        raise RuntimeError("tf_gain_grad: shapes do not match nbands x G blocks")
    filt = None if filt is None else _c(filt)
    lib = _lib.load()
    B = Btot // nbands
    if partial:
        parts = torch.empty((Btot * G, lib.gfdn_tf_gain_chunks(K)), dtype=_f32, device=gH.device)
        _lib.check(lib.gfdn_tf_gain_grad(K, nbands, G, B, _p(Tsave), _p(filt), K, _p(gH), K, None, _p(parts), _stream()),
                   "gfdn_tf_gain_grad")
        return parts
    grgain = torch.empty((Btot, G), dtype=_f32, device=gH.device) if grgain is None else grgain
    if work is None:
        work = _work(lib.gfdn_tf_gain_grad_work_bytes(K, nbands, G, B), gH.device)
    _lib.check(lib.gfdn_tf_gain_grad(K, nbands, G, B, _p(Tsave), _p(filt), K, _p(gH), K, _p(grgain), _p(work),
                                     _stream()), "gfdn_tf_gain_grad")
    return grgain


def tf_compose_bwd(turns, logr, coef, delays, nper: int, rgain, gH, Tsave, filt=None, nbands: int = 1, work=None,
                   partial: bool = False, tscale=None, gain_fold: bool = False):
    """-> grec (nbands*G, 32): gradient records of the scaled records; ``Tsave``: the forward's saved group transfer
    functions (tf_compose_fwd(save_T=True)).  ``partial``: skip the sum over the workgroups' partial rows and return
    them (nbands*G, 32, parts) for ``tf_param_grads``, which sums them itself.  ``tscale`` (nbands*G,): ``Tsave`` holds the
    UNSCALED functions and T' = tscale T is formed where they are read; ``gain_fold`` (with tscale): the scale sits in the
    receiver gains (tf_energy(gains=)) and gH is the gradient w.r.t. the UNSCALED group responses."""
    _need_gpu(turns, coef, rgain, gH, Tsave)
    if tscale is not None and (tscale.dtype != _f32 or not tscale.is_contiguous() or tscale.numel() != coef.shape[0]):
        raise RuntimeError("tf_compose_bwd: tscale must be a contiguous float32 vector with one factor per block")
    coef, delays, rgain, gH, Tsave = _f(coef), _f(delays), _f(rgain), _c(gH), _c(Tsave)
    K = turns.numel()
    Btot, G = rgain.shape
    if coef.shape[0] != nbands * G or Btot % nbands or tuple(gH.shape) != (Btot, K) \
            or tuple(Tsave.shape) != (nbands * G, K):
        raise RuntimeError("tf_compose_bwd: shapes do not match nbands x G blocks")
    filt = None if filt is None else _c(filt)
    lib = _lib.load()
    if partial:
        grec = None
        work = torch.empty((nbands * G, 32, lib.gfdn_tf_compose_parts(K)), dtype=_f32, device=coef.device)
    else:
        grec = torch.empty((nbands * G, 32), dtype=_f32, device=coef.device)
        if work is None:
            work = _work(lib.gfdn_tf_compose_bwd_work_bytes(K, nbands, G), coef.device)
    _lib.check(lib.gfdn_tf_compose_bwd(_p(turns), _p(logr), K, nbands, G, nper, _p(coef), _p(delays), _p(Tsave),
                                       _p(tscale), _p(rgain), Btot // nbands, _p(filt), K, _p(gH), K, _p(grec), _p(work),
                                       int(bool(gain_fold)), _stream()), "gfdn_tf_compose_bwd")
    return work if partial else grec


def tf_param_grads(A0, ig0, grec0, b, c, M, A1=None, ig1=None, grec1=None, gQ=None, Q=None, gb=None, gc=None, gM=None):
    """Tail of the block-transfer-function backward in one launch: ``grec0`` (nblk, 32) summed records or
    (nblk, 32, parts) partial rows (tf_compose_bwd(partial=True)) [+ ``grec1`` (nblk, 32) of a second record set on
    ``A1``] -> (gM, gb, gc) with the orthogonal parameterisation's adjoint folded in (``gQ``: gradient reaching Q
    directly, ``Q``: the forward's)."""
    _need_gpu(A0, grec0, b, c, M)
    A0, b, c, grec0, M = _f(A0), _f(b).reshape(-1), _f(c).reshape(-1), _f(grec0), _f(M)
    nblk, n, _ = A0.shape
    nparts0 = 1 if grec0.dim() == 2 else grec0.shape[2]
    if tuple(grec0.shape[:2]) != (nblk, 32) or (grec1 is not None and tuple(grec1.shape) != (nblk, 32)) \
            or tuple(M.shape) != (nblk, n, n):
        raise RuntimeError("tf_param_grads: records must be (nblk, 32[, parts]), M (nblk, n, n)")
    dev = A0.device
    ig0 = None if ig0 is None else _f(ig0).reshape(-1)
    if A1 is not None:
        A1, grec1 = _f(A1), _f(grec1)
        ig1 = None if ig1 is None else _f(ig1).reshape(-1)
    gb = torch.empty(nblk * n, dtype=_f32, device=dev) if gb is None else gb
    gc = torch.empty(nblk * n, dtype=_f32, device=dev) if gc is None else gc
    gM = torch.empty_like(M) if gM is None else gM
    _lib.check(_lib.load().gfdn_tf_param_grads(_p(A0), _p(ig0), _p(grec0), nparts0, _p(A1), _p(ig1), _p(grec1), _p(b),
                                               _p(c), nblk, n, _p(M), _p(None if gQ is None else _f(gQ)),
                                               _p(None if Q is None else _f(Q)), _p(gb), _p(gc), _p(gM), _stream()),
               "gfdn_tf_param_grads")
    return gM, gb, gc


def tf_tail(QQ, ig, grec0, grec1, b, c, M, gQ, Q, gb, gc, gM, opt, offM: int, offb: int, offc: int, Q_next, QQ_next,
            coef_next, coef_sub_next):
    """tf_param_grads + Adam on the blocks' own entries of M, b, c + tf_ortho_coefs of the updated blocks in ONE launch
    (csrc/blocktf.hip k_tf_tail).  ``grec0`` (nblk, 32, parts) partial rows of the records pass, ``grec1`` (nblk, 32) the
    raw blocks' summed records; ``b``, ``c``, ``M``: the flat parameter buffer's views of the leaves (updated in place);
    ``opt``: the FlatAdam whose buffers hold them at the element offsets ``offM`` / ``offb`` / ``offc``; the ``*_next``
    arrays receive the next step's Q, QQ and record sets (they may be the arrays this step read)."""
    _need_gpu(QQ, grec0, b, c, M)
    nblk, n, _ = QQ.shape
    nparts0 = 1 if grec0.dim() == 2 else grec0.shape[2]
    ig = None if ig is None else _f(ig).reshape(-1)
    for t in (QQ, grec0, grec1, b, c, M, gQ, Q, gb, gc, gM, Q_next, QQ_next, coef_next, coef_sub_next):
        if t is not None and (t.dtype != _f32 or not t.is_contiguous()):
            raise RuntimeError("tf_tail: contiguous float32 tensors expected")
    if tuple(grec0.shape[:2]) != (nblk, 32) or tuple(grec1.shape) != (nblk, 32) or M.numel() != nblk * n * n \
            or b.numel() != nblk * n or c.numel() != nblk * n:
        raise RuntimeError("tf_tail: records must be (nblk, 32[, parts]), M (nblk, n, n), b / c (nblk n)")
    fp = opt.flat_param
    for off, t in ((offM, M), (offb, b), (offc, c)):
        if fp.data_ptr() + 4 * off != t.data_ptr():
            raise RuntimeError("tf_tail: the leaves must be the flat buffer's views at the stated offsets")
    b1, b2 = opt.defaults['betas']
    _lib.check(_lib.load().gfdn_tf_tail(_p(QQ), _p(ig), _p(grec0), nparts0, _p(M), _p(grec1), _p(b), _p(c), nblk, n, _p(M),
                                        _p(gQ), _p(Q), _p(gb), _p(gc), _p(gM), _p(fp), _p(opt.exp_avg), _p(opt.exp_avg_sq),
                                        _p(opt.seg), _p(opt.lr_seg), _p(opt.step_count), _p(opt._block_counter), int(offM),
                                        int(offb), int(offc), float(b1), float(b2), float(opt.defaults['eps']),
                                        _p(Q_next), _p(QQ_next), _p(coef_next), _p(coef_sub_next), _stream()),
               "gfdn_tf_tail")


def tf_ortho_coefs(M, ig, b, c, sub: bool = True, out=None):
    """Head of the block-transfer-function step in one launch: (Q, QQ, coef of (QQ, 1 / gamma), coef_sub of (M, 1) or
    None) -- ``ortho_fwd`` + ``tf_coefs2`` with the same numbers.  ``out`` = (Q, QQ, coef, coef_sub): where to write them."""
    _need_gpu(M, b, c)
    M, b, c = _f(M), _f(b).reshape(-1), _f(c).reshape(-1)
    nblk, n, _ = M.shape
    ig = None if ig is None else _f(ig).reshape(-1)
    if out is not None:
        Q, QQ, coef, coef_sub = out
    else:
        Q, QQ = torch.empty_like(M), torch.empty_like(M)
        coef = torch.empty((nblk, 32), dtype=_f32, device=M.device)
        coef_sub = torch.empty((nblk, 32), dtype=_f32, device=M.device) if sub else None
    _lib.check(_lib.load().gfdn_tf_ortho_coefs(_p(M), _p(ig), _p(b), _p(c), nblk, n, _p(Q), _p(QQ), _p(coef),
                                               _p(coef_sub), _stream()), "gfdn_tf_ortho_coefs")
    return Q, QQ, coef, coef_sub


# ---- blocks of 5..8 lines: polynomial form on the matrix cores (csrc/blocktf8.hip) ---------------------------------------
def tf8_coefs(A0, ig0, b, c, A1=None, ig1=None):
    """Coefficient records (nblk, 9, 256) float32 of (A0, 1 / gamma0) [and (A1, 1 / gamma1)] sharing b: the determinant
    polynomial and the numerators of y = X^-1 b (see the kernel file; c is read by the passes)."""
    _need_gpu(A0, b, c)
    A0, b, c = _f(A0), _f(b).reshape(-1), _f(c).reshape(-1)
    nblk, n, _ = A0.shape
    if b.numel() != nblk * n or c.numel() != nblk * n:
        raise RuntimeError("tf8_coefs: b, c must hold nblk * n gains")
    ig0 = None if ig0 is None else _f(ig0).reshape(-1)
    c0 = torch.empty((nblk, 9, 256), dtype=_f32, device=A0.device)
    c1 = None
    if A1 is not None:
        A1 = _f(A1)
        ig1 = None if ig1 is None else _f(ig1).reshape(-1)
        c1 = torch.empty((nblk, 9, 256), dtype=_f32, device=A0.device)
    _lib.check(_lib.load().gfdn_tf8_coefs(_p(A0), _p(ig0), _p(c0), _p(A1), _p(ig1), _p(c1), _p(b), _p(c), nblk, n,
                                          _stream()), "gfdn_tf8_coefs")
    return c0, c1


def tf8_parts(K: int) -> int:
    return _lib.load().gfdn_tf8_parts(int(K))


def tf8_energy(turns, coef, delays, nper: int, b, c, want_energy: bool = False, dturn: float = 0.0):
    """normalize (trainer.py:317-332) on the records ``coef`` of the raw sub-FDN blocks: E = mean_k |sum_i c_i y_i|^2,
    -> (energy or None, scale = E^(-1/2)); b, c (float32, contiguous) are divided by E^(1/4) IN PLACE."""
    _need_gpu(turns, coef, delays, b, c)
    coef, delays = _f(coef), _f(delays)
    nblk, K = coef.shape[0], turns.numel()
    for t in (b, c):
        if t.dtype != _f32 or not t.is_contiguous() or t.numel() != nblk * nper:
            raise RuntimeError("tf8_energy: gains must be contiguous float32 of nblk * nper elements")
    lib = _lib.load()
    energy = torch.empty(nblk, dtype=_f32, device=coef.device) if want_energy else None
    scale = torch.empty(nblk, dtype=_f32, device=coef.device)
    work = torch.empty(nblk * lib.gfdn_tf8_parts(K), dtype=_f32, device=coef.device)
    _lib.check(lib.gfdn_tf8_energy(_p(turns), K, nblk, nper, _p(coef), _p(delays), _p(b), _p(c), _p(energy), _p(scale),
                                   _p(work), float(dturn), _stream()), "gfdn_tf8_energy")
    return energy, scale


def tf8_tsave(turns, coef, delays, nper: int, c, scale, nbands: int, G: int, quad: bool = True, filt=None,
              want_H: bool = False, hslot=None):
    """Scaled group transfer functions T' (nbands * G, K) complex64 [+ Tquad (nbands, K, 4)] from the records.
    ``want_H``: returns (Ts, Tq, Hg, Dinv) with Hg = T' filt (``filt`` (nbands, K) complex64 or None) and Dinv = 1 / Q per
    bin (what tf8_compose_bwd on the same grid takes back with Ts), written by the same launch.  ``hslot`` (K,) int32
    (tfp_slot_of_bin): the grid is in BIN order, Hg (and ``filt``) in the slot order of the odd-length transform -- column
    hslot[k] & 0x7fffffff, conjugated where bit 31 is set."""
    _need_gpu(turns, coef, delays, c)
    coef, delays, c = _f(coef), _f(delays), _f(c).reshape(-1)
    K = turns.numel()
    if coef.shape[0] != nbands * G:
        raise RuntimeError("tf8_tsave: records do not match nbands x G blocks")
    scale = None if scale is None else _f(scale)
    filt = None if filt is None else _c(filt)
    if filt is not None and filt.numel() != nbands * K:
        raise RuntimeError("tf8_tsave: filt must hold K bins per band")
    Ts = torch.empty((nbands * G, K), dtype=_c64, device=coef.device)
    Tq = torch.empty((nbands, K, 4), dtype=_c64, device=coef.device) if quad else None
    Hg = torch.empty((nbands * G, K), dtype=_c64, device=coef.device) if want_H else None
    Dinv = torch.empty((nbands * G, K), dtype=_c64, device=coef.device) if want_H else None
    _lib.check(_lib.load().gfdn_tf8_tsave(_p(turns), K, nbands, G, nper, _p(coef), _p(delays), _p(c), _p(scale), _p(Ts),
                                          _p(Tq), _p(filt if want_H else None), K, _p(Hg), _p(Dinv),
                                          _p(hslot if want_H else None), _stream()), "gfdn_tf8_tsave")
    return (Ts, Tq, Hg, Dinv) if want_H else (Ts, Tq)


def tf8_colorless(turns, coef, delays, nper: int, c, scale, asym: bool, gscale: float, dturn: float = 0.0):
    """Colorless pass on the records of the raw sub-FDN blocks -> (part (nblk, 512, parts) gradient records, loss (nblk,))."""
    _need_gpu(turns, coef, delays, c)
    coef, delays, c = _f(coef), _f(delays), _f(c).reshape(-1)
    nblk, K = coef.shape[0], turns.numel()
    lib = _lib.load()
    parts = lib.gfdn_tf8_parts(K)
    part = torch.empty((nblk, 512, parts), dtype=_f32, device=coef.device)
    lossp = torch.empty((nblk, parts), dtype=_f32, device=coef.device)
    loss = torch.empty(nblk, dtype=_f32, device=coef.device)
    _lib.check(lib.gfdn_tf8_colorless(_p(turns), K, nblk, nper, _p(coef), _p(delays), _p(c), _p(None if scale is None else _f(scale)),
                                      int(asym), float(gscale), _p(part), _p(lossp), _p(loss), float(dturn), _stream()),
               "gfdn_tf8_colorless")
    return part, loss


def tf8_compose_bwd(turns, coef, delays, nper: int, c, scale, rgain, gH, filt=None, nbands: int = 1, saved=None):
    """Gradient records (nbands * G, 512, parts) of the output stage from dL/dH (nbands * B, K).  ``saved`` = (Ts, Dinv) of
    tf8_tsave(want_H=True) on the SAME grid: the pass reads T' and 1 / Q instead of evaluating the polynomials again."""
    _need_gpu(turns, coef, delays, c, rgain, gH)
    coef, delays, c, rgain, gH = _f(coef), _f(delays), _f(c).reshape(-1), _f(rgain), _c(gH)
    K = turns.numel()
    Btot, G = rgain.shape
    if coef.shape[0] != nbands * G or Btot % nbands or gH.shape[0] != Btot or gH.shape[1] < K:
        raise RuntimeError("tf8_compose_bwd: shapes do not match nbands x G blocks")
    filt = None if filt is None else _c(filt)
    if filt is not None and filt.numel() != nbands * K:
        raise RuntimeError("tf8_compose_bwd: filt must hold K bins per band")
    lib = _lib.load()
    part = torch.empty((nbands * G, 512, lib.gfdn_tf8_parts(K)), dtype=_f32, device=coef.device)
    Ts = Dinv = None
    if saved is not None:
        Ts, Dinv = saved
        for t in (Ts, Dinv):
            if t.dtype != _c64 or not t.is_contiguous() or tuple(t.shape) != (nbands * G, K):
                raise RuntimeError("tf8_compose_bwd: saved = (Ts, Dinv), contiguous complex64 (nbands * G, K)")
    _lib.check(lib.gfdn_tf8_compose_bwd(_p(turns), K, nbands, G, nper, _p(coef), _p(delays), _p(c),
                                        _p(None if scale is None else _f(scale)), _p(rgain), Btot // nbands, _p(filt), K,
                                        _p(gH), gH.stride(0), _p(Ts), _p(Dinv), _p(part), _stream()), "gfdn_tf8_compose_bwd")
    return part


def tf8_param_grads(A0, ig0, part0, b, c, M, A1=None, ig1=None, part1=None, gQ=None, Q=None, gb=None, gc=None, gM=None):
    """Tail of the 8-line step: gradient records of set 0 (A0 = Q Q, 1 / gamma0) [and set 1 (A1 = raw M)] ->
    (gM, gb, gc) w.r.t. the CURRENT gains b, c, with the orthogonal parameterisation's adjoint folded in."""
    _need_gpu(A0, part0, b, c, M)
    A0, M, b, c = _f(A0), _f(M), _f(b).reshape(-1), _f(c).reshape(-1)
    nblk, n, _ = M.shape
    if part0.dim() != 3 or tuple(part0.shape[:2]) != (nblk, 512) or \
            (part1 is not None and (A1 is None or tuple(part1.shape[:2]) != (nblk, 512))):
        raise RuntimeError("tf8_param_grads: records must be (nblk, 512, parts)")
    dev = M.device
    ig0 = None if ig0 is None else _f(ig0).reshape(-1)
    ig1 = None if ig1 is None else _f(ig1).reshape(-1)
    gb = torch.empty(nblk * n, dtype=_f32, device=dev) if gb is None else gb
    gc = torch.empty(nblk * n, dtype=_f32, device=dev) if gc is None else gc
    gM = torch.empty_like(M) if gM is None else gM
    lib = _lib.load()
    work = torch.empty(lib.gfdn_tf8_param_grads_work_bytes(nblk) // 4, dtype=_f32, device=dev)
    _lib.check(lib.gfdn_tf8_param_grads(_p(A0), _p(ig0), _p(_f(part0)), part0.shape[2],
                                        _p(None if A1 is None else _f(A1)), _p(ig1),
                                        _p(None if part1 is None else _f(part1)), 0 if part1 is None else part1.shape[2],
                                        _p(b), _p(c), nblk, n, _p(M), _p(None if gQ is None else _f(gQ)),
                                        _p(None if Q is None else _f(Q)), _p(gb), _p(gc), _p(gM), _p(work), _stream()),
               "gfdn_tf8_param_grads")
    return gM, gb, gc


# ---- the 8-line block transfer functions on the rfftfreq grid by fast transforms (csrc/polyfft.hip) ----
_tfp_plans = {}
_tfp_slots = {}


def tfp_plan(delays, nper: int, nfft: int):
    """Samples per coefficient sequence (max degree + 1, rounded up to 256, <= nfft) when every delay length is a
    non-negative integer (the reference's are: config.py:131-140), else None.  Reads the device once per delays tensor."""
    key = (delays.data_ptr(), delays._version, delays.numel(), int(nper), int(nfft), str(delays.device))
    if key not in _tfp_plans:
        if delays.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("tfp_plan: first use of this delays tensor inside a stream capture; run the step once before")
        d = delays.detach().to(torch.float64).cpu().reshape(-1, nper)
        T = None
        if nfft >= 16 and nfft & (nfft - 1) == 0 and bool((d == d.round()).all()) and bool((d >= 0).all()):
            T = int(min(nfft, ((int(d.sum(1).max().item()) + 256) // 256) * 256))
        while len(_tfp_plans) >= 64:
            _tfp_plans.pop(next(iter(_tfp_plans)))
        # (the entry keeps the tensor alive: the key is its ADDRESS, and a freed tensor's block handed to another delays
        # tensor of the same size would be a false hit -- rows too short for the new degrees)
        _tfp_plans[key] = (T, delays)
    return _tfp_plans[key][0]


def tfp_slot_of_bin(n: int, device):
    """(Ku,) int32: column of bin k in the slot-ordered spectrum of irfft(X, n), n odd (column 0 = bin 0, 1 + s = slot s of
    irfft_slot_order), bit 31 set where the slot holds the conjugate -- or None when the length has no slot order."""
    key = (int(n), str(device))
    if key not in _tfp_slots:
        order = irfft_slot_order(n, device)
        if order is None:
            _tfp_slots[key] = None
        else:
            bins, conj = order[0].cpu(), order[1].cpu()
            col = torch.zeros((int(n) + 1) // 2, dtype=torch.int64)
            col[bins] = torch.arange(1, bins.numel() + 1, dtype=torch.int64) | (conj.to(torch.int64) << 31)
            col = torch.where(col >= 2 ** 31, col - 2 ** 32, col)
            _tfp_slots[key] = col.to(torch.int32).to(device)
    return _tfp_slots[key]


def tfp_forward(coef, delays, c, nper: int, nfft: int, T: int) -> torch.Tensor:
    """X (2 nblk, nfft / 2 + 1) complex64 = rfft of the coefficient sequences [Q | P] of one record set: Q(z_k) =
    conj(X[blk][k]), P(z_k) = conj(X[nblk + blk][k]) on z_k = e^{2 pi i k / nfft}.  P is formed with the gains ``c`` as they
    are NOW (before normalize's rescale)."""
    _need_gpu(coef, delays, c)
    coef, delays, c = _f(coef), _f(delays), _f(c).reshape(-1)
    nblk = coef.shape[0]
    K = nfft // 2 + 1
    lib = _lib.load()
    seq = torch.empty((2 * nblk, T), dtype=_f32, device=coef.device)
    X = torch.empty((2 * nblk, K), dtype=_c64, device=coef.device)
    work = _work(lib.gfdn_irfft_pow2_work_bytes(nfft, 2 * nblk), coef.device)
    nsub = coef.shape[-1]                 # (256: the records of tf8_coefs, 512: of tf9_coefs)
    _lib.check(lib.gfdn_tfp_forward(nfft, nblk, nper, nsub, _p(coef), _p(delays), _p(c), T, _p(seq), _p(X), K, _p(work),
                                    _stream()), "gfdn_tfp_forward")
    return X


def tfp_energy(Xq, Xp, nper: int, b, c, want_energy: bool = False, gains=None, G: int = 0):
    """normalize (trainer.py:317-332) on the transformed sequences of the raw sub-FDN blocks (rows Xq, Xp of tfp_forward):
    -> (energy or None, scale = E^(-1/2)); b, c (float32, contiguous) are divided by E^(1/4) IN PLACE.  ``gains``: as
    tf_energy's (returns a third element, the scaled gains)."""
    _need_gpu(Xq, Xp, b, c)
    nblk, K = Xq.shape
    for t in (b, c):
        if t.dtype != _f32 or not t.is_contiguous() or t.numel() != nblk * nper:
            raise RuntimeError("tfp_energy: gains must be contiguous float32 of nblk * nper elements")
    if Xq.stride(0) != Xp.stride(0) or Xq.stride(1) != 1 or Xp.stride(1) != 1:
        raise RuntimeError("tfp_energy: rows of one tfp_forward result")
    lib = _lib.load()
    energy = torch.empty(nblk, dtype=_f32, device=Xq.device) if want_energy else None
    scale = torch.empty(nblk, dtype=_f32, device=Xq.device)
    work = torch.empty(nblk * lib.gfdn_tfp_parts(), dtype=_f32, device=Xq.device)
    gs = None
    if gains is not None:
        if G <= 0 or nblk % G or gains.dtype != _f32 or not gains.is_contiguous() or gains.shape[1] != G \
                or gains.shape[0] % (nblk // G):
            raise RuntimeError("tfp_energy: gains (bands * Bper, G) float32 contiguous with nblk = bands * G")
        gs = torch.empty_like(gains)
    _lib.check(lib.gfdn_tfp_energy(_p(Xq), _p(Xp), Xq.stride(0), K, nblk, nper, _p(b), _p(c), _p(energy), _p(scale), _p(work),
                                   _p(gains), _p(gs), 0 if gains is None else gains.shape[0] // (nblk // G), int(G), _stream()),
               "gfdn_tfp_energy")
    return (energy, scale) if gains is None else (energy, scale, gs)


def tfp_colorless(Xq, Xp, nfft: int, nper: int, delays, scale, asym: bool, gscale: float, T: Optional[int] = None):
    """tf8_colorless on the transformed sequences -> (part (nblk, 512, 1) gradient records, loss (nblk,)).  ``T``: the
    samples per sequence of tfp_forward (tfp_plan): what the gather reads of the inverse transforms (default: all)."""
    _need_gpu(Xq, Xp, delays)
    nblk, K = Xq.shape
    if K != nfft // 2 + 1 or Xq.stride(0) != Xp.stride(0):
        raise RuntimeError("tfp_colorless: rows of one tfp_forward result on nfft / 2 + 1 bins")
    lib = _lib.load()
    dev = Xq.device
    UV = torch.empty((2 * nblk, K), dtype=_c64, device=dev)
    x = torch.empty((2 * nblk, nfft), dtype=_f32, device=dev)
    work = _work(lib.gfdn_irfft_pow2_work_bytes(nfft, 2 * nblk), dev)
    part = torch.empty((nblk, 512, 1), dtype=_f32, device=dev)
    lossp = torch.empty(nblk * lib.gfdn_tfp_parts(), dtype=_f32, device=dev)
    loss = torch.empty(nblk, dtype=_f32, device=dev)
    _lib.check(lib.gfdn_tfp_colorless(_p(Xq), _p(Xp), Xq.stride(0), nfft, nblk, nper, _p(_f(delays)),
                                      _p(None if scale is None else _f(scale)), int(asym), float(gscale),
                                      int(nfft if T is None else T), _p(UV), _p(x), nfft, _p(work), _p(part), _p(lossp), _p(loss),
                                      _stream()), "gfdn_tfp_colorless")
    return part, loss


def tfp_compose_bwd(nfft: int, nbands: int, G: int, nper: int, delays, Ku: int, slot_of_bin, gH, filt, Tnat, Dnat, tscale=None,
                    gain_fold: bool = False, T: Optional[int] = None):
    """Gradient records (nbands * G, 512, 1) of the damped blocks from gH (nbands * G, >= Ku) = dL/d(T'_g filt) on the slot
    order (the linear step's adjoint transform output); Tnat, Dnat (nbands * G, Ku): tf8_tsave's Ts, Dinv on the bins
    0 .. Ku - 1 in bin order (``hslot``); ``tscale``: Tnat holds the unscaled functions, T' = tscale Tnat."""
    _need_gpu(gH, Tnat, Dnat, delays)
    nblk = nbands * G
    gH = _c(gH)
    filt = None if filt is None else _c(filt)
    if gH.shape[0] != nblk or gH.shape[1] < Ku or tuple(Tnat.shape) != (nblk, Ku) or tuple(Dnat.shape) != (nblk, Ku):
        raise RuntimeError("tfp_compose_bwd: shapes do not match nbands x G blocks on Ku bins")
    if filt is not None and filt.numel() != nbands * Ku:
        raise RuntimeError("tfp_compose_bwd: filt must hold Ku slots per band")
    lib = _lib.load()
    dev = gH.device
    K = nfft // 2 + 1
    UV = torch.empty((2 * nblk, K), dtype=_c64, device=dev)
    x = torch.empty((2 * nblk, nfft), dtype=_f32, device=dev)
    work = _work(lib.gfdn_irfft_pow2_work_bytes(nfft, 2 * nblk), dev)
    part = torch.empty((nblk, 512, 1), dtype=_f32, device=dev)
    _lib.check(lib.gfdn_tfp_compose_bwd(nfft, nbands, G, nper, _p(_f(delays)), Ku, _p(slot_of_bin), _p(gH), gH.stride(0), _p(filt),
                                        Ku, _p(Tnat), _p(Dnat), _p(None if tscale is None else _f(tscale)),
                                        int(bool(gain_fold)), int(nfft if T is None else T), _p(UV), K, _p(x), nfft, _p(work),
                                        _p(part), _stream()),
               "gfdn_tfp_compose_bwd")
    return part


def tf9_coefs(A, ig, b):
    """Coefficient records (nblk, 10, 512) float32 of blocks of up to nine lines: the determinant polynomial of
    X = D Gamma^-1 - A and the nine numerators of y = X^-1 b (csrc/blocktf9.hip)."""
    _need_gpu(A, b)
    A, b = _f(A), _f(b).reshape(-1)
    nblk, n, _ = A.shape
    if n > 9 or b.numel() != nblk * n:
        raise RuntimeError("tf9_coefs: blocks of at most nine lines, b (nblk n)")
    ig = None if ig is None else _f(ig).reshape(-1)
    coef = torch.empty((nblk, 10, 512), dtype=_f32, device=A.device)
    _lib.check(_lib.load().gfdn_tf9_coefs(_p(A), _p(ig), _p(b), nblk, n, _p(coef), _stream()), "gfdn_tf9_coefs")
    return coef


def tf9_rec_grads(A, ig, grec, b, c):
    """Gradient records (nblk, 1024[, 1]) of blocks of up to nine lines -> (dL/dA (nblk, n, n), dL/db, dL/dc (nblk n))."""
    _need_gpu(A, grec, b, c)
    A, b, c, grec = _f(A), _f(b).reshape(-1), _f(c).reshape(-1), _f(grec)
    nblk, n, _ = A.shape
    if grec.numel() != nblk * 1024:
        raise RuntimeError("tf9_rec_grads: records must be (nblk, 1024)")
    ig = None if ig is None else _f(ig).reshape(-1)
    lib = _lib.load()
    gA = torch.empty_like(A)
    gb = torch.empty(nblk * n, dtype=_f32, device=A.device)
    gc = torch.empty(nblk * n, dtype=_f32, device=A.device)
    work = _work(lib.gfdn_tf9_rec_grads_work_bytes(nblk), A.device)
    _lib.check(lib.gfdn_tf9_rec_grads(_p(A), _p(ig), _p(grec), _p(b), _p(c), nblk, n, _p(gA), _p(gb), _p(gc), _p(work),
                                      _stream()), "gfdn_tf9_rec_grads")
    return gA, gb, gc


def tfp_ratio_fwd(Xq, Xp):
    """(T, Dinv) (nblk, K) complex64 = (P / Q, 1 / Q) on the grid from the transformed sequences (rows of tfp_forward)."""
    _need_gpu(Xq, Xp)
    nblk, K = Xq.shape
    if Xq.stride(0) != Xp.stride(0) or Xq.stride(1) != 1:
        raise RuntimeError("tfp_ratio_fwd: rows of one tfp_forward result")
    T = torch.empty((nblk, K), dtype=_c64, device=Xq.device)
    D = torch.empty((nblk, K), dtype=_c64, device=Xq.device)
    _lib.check(_lib.load().gfdn_tfp_ratio_fwd(_p(Xq), _p(Xp), Xq.stride(0), K, nblk, _p(T), _p(D), _stream()), "gfdn_tfp_ratio_fwd")
    return T, D


def tfp_ratio_bwd(nfft: int, nper: int, delays, gT, T, Dinv, nsub: int = 512, T_seq: Optional[int] = None):
    """Gradient records (nblk, 2 nsub) (nsub = 256: subsets of tf8_coefs' records, 512: of tf9_coefs') of T = P / Q from gT =
    dL/dT (nblk, K), dL = sum_k Re(conj(gT_k) dT_k): two inverse real transforms and a gather (csrc/polyfft.hip)."""
    _need_gpu(gT, T, Dinv, delays)
    gT = _c(gT)
    nblk, K = T.shape
    if K != nfft // 2 + 1 or tuple(gT.shape) != (nblk, K):
        raise RuntimeError("tfp_ratio_bwd: gT and T on the nfft / 2 + 1 bins of the grid")
    lib = _lib.load()
    dev = T.device
    UV = torch.empty((2 * nblk, K), dtype=_c64, device=dev)
    x = torch.empty((2 * nblk, nfft), dtype=_f32, device=dev)
    work = _work(lib.gfdn_irfft_pow2_work_bytes(nfft, 2 * nblk), dev)
    part = torch.empty((nblk, 2 * nsub), dtype=_f32, device=dev)
    _lib.check(lib.gfdn_tfp_ratio_bwd(nfft, nblk, nper, int(nsub), _p(_f(delays)), _p(gT), K, _p(T), _p(Dinv),
                                      int(nfft if T_seq is None else T_seq), _p(UV), K, _p(x), nfft,
                                      _p(work), _p(part), _stream()), "gfdn_tfp_ratio_bwd")
    return part


def tf8_tail(QQ, ig, part0, part1, b, c, M, gQ, Q, gb, gc, gM, opt, offM: int, offb: int, offc: int, Q_next, QQ_next, c_next):
    """tf8_param_grads + Adam on the blocks' own entries of M, b, c + the next step's Q, Q Q and output-gain snapshot in ONE
    launch behind the cofactor maps (csrc/blocktf8.hip k_tf8_tail).  ``part0`` / ``part1`` (nblk, 512, parts): gradient
    records of the damped blocks (A0 = Q Q, 1 / gamma) and of the raw blocks (A1 = M); ``b``, ``c``, ``M``: the flat parameter
    buffer's views of the leaves (updated in place); ``opt``: the FlatAdam that holds them at the element offsets."""
    _need_gpu(QQ, part0, part1, b, c, M)
    nblk, n, _ = M.shape
    ig = None if ig is None else _f(ig).reshape(-1)
    for t in (QQ, part0, part1, b, c, M, gQ, Q, gb, gc, gM, Q_next, QQ_next, c_next):
        if t is not None and (t.dtype != _f32 or not t.is_contiguous()):
            raise RuntimeError("tf8_tail: contiguous float32 tensors expected")
    if part0.dim() != 3 or tuple(part0.shape[:2]) != (nblk, 512) or part1.dim() != 3 or tuple(part1.shape[:2]) != (nblk, 512) \
            or b.numel() != nblk * n or c.numel() != nblk * n or c_next.numel() != nblk * n:
        raise RuntimeError("tf8_tail: records must be (nblk, 512, parts), b / c / c_next (nblk n)")
    fp = opt.flat_param
    for off, t in ((offM, M), (offb, b), (offc, c)):
        if fp.data_ptr() + 4 * off != t.data_ptr():
            raise RuntimeError("tf8_tail: the leaves must be the flat buffer's views at the stated offsets")
    lib = _lib.load()
    work = torch.empty(lib.gfdn_tf8_param_grads_work_bytes(nblk) // 4, dtype=_f32, device=M.device)
    b1, b2 = opt.defaults['betas']
    _lib.check(lib.gfdn_tf8_tail(_p(QQ), _p(ig), _p(part0), part0.shape[2], _p(M), _p(part1), part1.shape[2], _p(b), _p(c), nblk,
                                 n, _p(M), _p(gQ), _p(Q), _p(gb), _p(gc), _p(gM), _p(work), _p(fp), _p(opt.exp_avg),
                                 _p(opt.exp_avg_sq), _p(opt.seg), _p(opt.lr_seg), _p(opt.step_count), _p(opt._block_counter),
                                 int(offM), int(offb), int(offc), float(b1), float(b2), float(opt.defaults['eps']),
                                 _p(Q_next), _p(QQ_next), _p(c_next), _stream()), "gfdn_tf8_tail")


def tf_coefs_bwd(A0, ig0, grec0, b, c, A1=None, ig1=None, grec1=None, gA0=None, gA1=None, gb=None, gc=None):
    """Gradient records -> (gA0, gA1 or None, gb, gc); b, c: the gains the records' gradients refer to."""
    _need_gpu(A0, grec0, b, c)
    A0, b, c, grec0 = _f(A0), _f(b).reshape(-1), _f(c).reshape(-1), _f(grec0)
    nblk, n, _ = A0.shape
    ig0 = None if ig0 is None else _f(ig0).reshape(-1)
    dev = A0.device
    if tuple(grec0.shape) != (nblk, 32) or (grec1 is not None and tuple(grec1.shape) != (nblk, 32)):
        raise RuntimeError("tf_coefs_bwd: gradient records must be (nblk, 32)")
    gA0 = torch.empty_like(A0) if gA0 is None else gA0
    gb = torch.empty(nblk * n, dtype=_f32, device=dev) if gb is None else gb
    gc = torch.empty(nblk * n, dtype=_f32, device=dev) if gc is None else gc
    if A1 is not None:
        A1, grec1 = _f(A1), _f(grec1)
        ig1 = None if ig1 is None else _f(ig1).reshape(-1)
        gA1 = torch.empty_like(A1) if gA1 is None else gA1
    _lib.check(_lib.load().gfdn_tf_coefs_bwd(_p(A0), _p(ig0), _p(grec0), _p(A1), _p(ig1), _p(grec1), _p(b), _p(c),
                                             nblk, n, _p(gA0), _p(gA1), _p(gb), _p(gc), _stream()),
               "gfdn_tf_coefs_bwd")
    return gA0, gA1, gb, gc


def ortho_bwd_add(M, gQ, gQQ, Q, gM_add, out=None):
    """ortho_bwd with ``gM_add`` (G, n, n) added to the result; ``out``: where to write gM."""
    _need_gpu(M)
    M = _f(M)
    G, n, _ = M.shape
    gM = torch.empty_like(M) if out is None else out
    _lib.check(_lib.load().gfdn_ortho_bwd_add(_p(M), G, n, _p(None if gQ is None else _f(gQ)),
                                              _p(None if gQQ is None else _f(gQQ)), _p(None if Q is None else _f(Q)),
                                              _p(None if gM_add is None else _f(gM_add)), _p(gM), _stream()),
               "gfdn_ortho_bwd_add")
    return gM


# ------------------------------------------------------------------------------------------------
_blu_tables = {}


def bluestein_table(n: int, device) -> torch.Tensor:
    """Plan data for irfft(X, n), n odd (cached per (n, device))."""
    key = (int(n), str(device))
    t = _blu_tables.get(key)
    if t is None:
        lib = _lib.load()
        nbytes = lib.gfdn_bluestein_table_bytes(int(n))
        if nbytes == 0:
            raise RuntimeError(f"irfft_odd: n={n} must be odd and >= 3")
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _lib.check(lib.gfdn_bluestein_table_init(int(n), _p(t)), "gfdn_bluestein_table_init")
        _blu_tables[key] = t
    return t


_slot_orders = {}


def irfft_slot_order(n: int, device):
    """(bins (s,) int64, conj (s,) bool) on ``device`` for the slot-ordered spectrum of irfft(X, n) -- slot s holds
    X[bins[s]] (conjugated where conj[s]) -- or None when this length has no slot-order path."""
    key = (int(n), str(device))
    if key not in _slot_orders:
        import ctypes
        half = (int(n) - 1) // 2
        bins = (ctypes.c_int * max(half, 1))()
        conj = (ctypes.c_int * max(half, 1))()
        rc = _lib.load().gfdn_irfft_odd_slot_order(int(n), bins, conj) if n >= 3 and n % 2 == 1 else -2
        if rc == 0:
            _slot_orders[key] = (torch.tensor(list(bins), dtype=torch.int64, device=device),
                                 torch.tensor(list(conj), dtype=torch.bool, device=device))
        else:
            _slot_orders[key] = None
    return _slot_orders[key]


def irfft_odd_fwd(X, n: int, slots: bool = False, pairs: bool = False, oscale=None, before_last=None) -> torch.Tensor:
    """X (batch, >= (n+1)/2) c64 -> x (batch, n) float32 = torch.fft.irfft(X, n), n odd.
    ``slots``: X is in slot order (irfft_slot_order): X[:, 0] = bin 0, X[:, 1 + s] = slot s.
    ``pairs`` (with slots): two items per transform; returns x2 (ceil(batch / 2), n, 2) float32, item 2p in
    [..., 0], item 2p + 1 in [..., 1] (zeros for the missing partner of an odd batch).
    ``oscale`` (pairs only; (batch,) float32): x[item] = oscale[item] * irfft(X[item]), applied where the last pass stores;
    ``before_last``: callable run between the second and the last pass (e.g. the wait for the stream that produces oscale)."""
    _need_gpu(X)
    if oscale is not None and not pairs:
        raise RuntimeError("irfft_odd_fwd: oscale goes with pairs=True")
    if pairs:
        if not slots:
            raise RuntimeError("irfft_odd_fwd(pairs=True) takes slot-ordered spectra")
        X = _c(X)
        batch, ldx = X.shape
        lib = _lib.load()
        table = bluestein_table(n, X.device)
        x2 = torch.empty(((batch + 1) // 2, n, 2), dtype=_f32, device=X.device)
        work = _work(lib.gfdn_bluestein_work_bytes(n, batch), X.device)
        if oscale is not None:
            if oscale.dtype != _f32 or not oscale.is_contiguous() or oscale.numel() != batch:
                raise RuntimeError("irfft_odd_fwd: oscale must be a contiguous float32 vector with one factor per item")
            args = (_p(table), n, _p(X), ldx, batch, _p(oscale), _p(x2), n, _p(work))
            if before_last is None:
                _lib.check(lib.gfdn_irfft_odd_pairs_fwd_scaled(*args, 7, _stream()), "gfdn_irfft_odd_pairs_fwd_scaled")
            else:
                _lib.check(lib.gfdn_irfft_odd_pairs_fwd_scaled(*args, 3, _stream()), "gfdn_irfft_odd_pairs_fwd_scaled")
                before_last()
                _lib.check(lib.gfdn_irfft_odd_pairs_fwd_scaled(*args, 4, _stream()), "gfdn_irfft_odd_pairs_fwd_scaled")
            return x2
        if kernel_timer.active and kernel_timer.watch in _BLU_STAGES:
            _staged_bluestein(lib, table, n, X, None, ldx, batch, x2, n, work, 0, 2)
            return x2
        _lib.check(lib.gfdn_irfft_odd_pairs_fwd(_p(table), n, _p(X), ldx, batch, _p(x2), n, _p(work), _stream()),
                   "gfdn_irfft_odd_pairs_fwd")
        return x2
    X = _c(X)
    batch, ldx = X.shape
    lib = _lib.load()
    table = bluestein_table(n, X.device)
    x = torch.empty((batch, n), dtype=_f32, device=X.device)
    work = _work(lib.gfdn_bluestein_work_bytes(n, batch), X.device)
    if kernel_timer.active and kernel_timer.watch in _BLU_STAGES:
        _staged_bluestein(lib, table, n, X, None, ldx, batch, x, n, work, 0, slots)
        return x
    fn = lib.gfdn_irfft_odd_slots_fwd if slots else lib.gfdn_irfft_odd_fwd
    _lib.check(fn(_p(table), n, _p(X), ldx, batch, _p(x), n, _p(work), _stream()), "gfdn_irfft_odd_fwd")
    return x


def irfft_odd_pairs_compose_fwd(direct, rows, T, rgain, filt, n: int, nbands: int) -> torch.Tensor:
    """x2 = irfft_odd_fwd(H, n, slots=True, pairs=True) for the output stage's H = (sum_g rgain T_g + direct[rows]) filt
    formed inside the transform's first pass (H is never stored).  direct (R, (n+1)/2) c64 slot-ordered store, rows
    (batch) int64 or None, T (nbands, (n+1)/2, 4) c64 (tf_compose_fwd(want_H=False)[0]), rgain (batch, G), filt
    (nbands, (n+1)/2) c64 or None.  Returns (x2, h0)."""
    _need_gpu(direct, T, rgain)
    direct, T, rgain = _c(direct), _c(T), _f(rgain)
    batch, G = rgain.shape
    half = (n + 1) // 2
    if direct.shape[1] != half or tuple(T.shape) != (nbands, half, 4) or batch % nbands:
        raise RuntimeError("irfft_odd_pairs_compose_fwd: shapes do not match nbands x G blocks / (n + 1) / 2 columns")
    filt = None if filt is None else _c(filt)
    rows = None if rows is None else _rows(rows, batch, direct.shape[0])
    lib = _lib.load()
    table = bluestein_table(n, direct.device)
    x2 = torch.empty((batch // 2, n, 2), dtype=_f32, device=direct.device)
    h0 = torch.empty(batch, dtype=_f32, device=direct.device)
    work = _work(lib.gfdn_bluestein_work_bytes(n, batch), direct.device)
    args = (_p(table), n, _p(direct), half, _p(rows), _p(T), half, _p(rgain), _p(filt), half, nbands, G, batch, _p(h0),
            _p(x2), n, _p(work))
    if kernel_timer.active and kernel_timer.watch in _BLU_STAGES:        # the three passes with events around one
        for name, stage in _BLU_STAGES.items():
            end = kernel_timer.bracket(name, batch)
            _lib.check(lib.gfdn_irfft_odd_pairs_compose_fwd(*args, stage, _stream()),
                       "gfdn_irfft_odd_pairs_compose_fwd[%s]" % name)
            if end is not None:
                end.record()
        return x2, h0
    _lib.check(lib.gfdn_irfft_odd_pairs_compose_fwd(*args, 7, _stream()), "gfdn_irfft_odd_pairs_compose_fwd")
    return x2, h0


# ---- the output stage in the time domain (csrc/linear.hip) -----------------------------------------------------------
def lin_combine_fwd(xd, rows, tau, rgain, nbands: int, n: int, tau_pairs: bool, out_pairs: bool):
    """x[b] = xd[rows[b]] + sum_g rgain[b][g] tau[band(b) G + g] over n time samples.  xd (R, >= n) float32 store of the
    transformed direct paths, tau the nbands*G transformed group responses -- (ceil(S / 2), n, 2) pair-interleaved
    (``tau_pairs``, what irfft_odd_fwd(pairs=True) returns) or (S, n); rgain (items, G).  Returns x2 (ceil(items / 2),
    n, 2) pair-interleaved (``out_pairs``) or x (items, n)."""
    _need_gpu(xd, tau, rgain)
    xd, tau, rgain = _f(xd), _f(tau), _f(rgain)
    items, G = rgain.shape
    S = nbands * G
    if items % nbands or xd.dim() != 2 or xd.shape[1] < n:
        raise RuntimeError("lin_combine_fwd: rgain (nbands * B, G), xd (R, >= n)")
    if tuple(tau.shape) != (((S + 1) // 2, n, 2) if tau_pairs else (S, n)):
        raise RuntimeError("lin_combine_fwd: tau does not hold nbands * G signals of n samples in the stated layout")
    rows = _rows(rows, items, xd.shape[0])
    if rows is None and xd.shape[0] != items:
        raise RuntimeError("lin_combine_fwd: xd must have one row per item (or pass rows)")
    x = torch.empty(((items + 1) // 2, n, 2) if out_pairs else (items, n), dtype=_f32, device=xd.device)
    _lib.check(_lib.load().gfdn_lin_combine_fwd(_p(xd), xd.stride(0), _p(rows), _p(tau), n, int(tau_pairs), _p(rgain),
                                                nbands, items // nbands, G, n, _p(x), n, int(out_pairs), _stream()),
               "gfdn_lin_combine_fwd")
    return x


_time_slots = {}


def irfft_slot_of_time(n: int, device):
    """(n,) int32 on ``device``: slot_of_time[t] = the convolution slot whose time index is t (t >= 1; entry 0 unused) --
    the inverse of gfdn_irfft_odd_time_slots -- or None when the length has no slot-order path."""
    key = (int(n), str(device))
    if key not in _time_slots:
        import ctypes
        times = (ctypes.c_int * max(int(n) - 1, 1))()
        rc = _lib.load().gfdn_irfft_odd_time_slots(int(n), times) if n >= 3 and n % 2 == 1 else -2
        if rc == 0:
            tt = torch.tensor(list(times), dtype=torch.int64)
            inv = torch.zeros(int(n), dtype=torch.int32)
            inv[tt] = torch.arange(int(n) - 1, dtype=torch.int32)
            _time_slots[key] = inv.to(device)
        else:
            _time_slots[key] = None
    return _time_slots[key]


def lin_gamma(gx, rgain, nbands: int, n: int, in_pairs: bool, out_pairs: bool, gxb=None, slot_of_time=None, base=None):
    """gamma[band G + g] = sum_{b in band} rgain[b][g] (gx[b] [+ gxb[b]]): the signals whose adjoint transform is
    dL/d(T_g filt).  gx: (ceil(items / 2), n, 2) pair-interleaved (``in_pairs``) or (items, n); result in the layout
    ``out_pairs`` asks for (what irfft_odd_pairs_bwd / irfft_odd_bwd take).  ``slot_of_time`` (irfft_slot_of_time): the
    pair-interleaved result in the adjoint transform's own order, for irfft_odd_pairs_bwd(tslots=True)."""
    _need_gpu(gx, rgain)
    gx, rgain = _f(gx), _f(rgain)
    items, G = rgain.shape
    S = nbands * G
    if items % nbands or tuple(gx.shape) != (((items + 1) // 2, n, 2) if in_pairs else (items, n)):
        raise RuntimeError("lin_gamma: gradient signals do not match rgain (nbands * B, G) in the stated layout")
    if gxb is not None:
        gxb = _f(gxb)
        if gxb.shape != gx.shape:
            raise RuntimeError("lin_gamma: gxb must have the shape of gx")
    gamma = torch.empty(((S + 1) // 2, n, 2) if out_pairs else (S, n), dtype=_f32, device=gx.device)
    if out_pairs and S % 2:
        gamma[-1].zero_()                  # (the missing partner of the last signal)
    if slot_of_time is not None and (not out_pairs or slot_of_time.dtype != torch.int32 or slot_of_time.numel() != n):
        raise RuntimeError("lin_gamma: slot_of_time is an int32 table of n entries for the pair-interleaved output")
    if base is not None:                   # (a part of the same signals already summed per group: time order, pairs)
        if not out_pairs or base.dtype != _f32 or not base.is_contiguous() or tuple(base.shape) != ((S + 1) // 2, n, 2):
            raise RuntimeError("lin_gamma: base must be pair-interleaved (ceil(S / 2), n, 2) float32")
    _lib.check(_lib.load().gfdn_lin_gamma(_p(gx), _p(gxb), n, int(in_pairs), _p(rgain), nbands, items // nbands, G, n,
                                          _p(gamma), n, int(out_pairs), _p(slot_of_time), _p(base), n, _stream()),
               "gfdn_lin_gamma")
    return gamma


def lin_gain_dots(gx, tau, nbands: int, items: int, G: int, n: int, in_pairs: bool, tau_pairs: bool, gxb=None, out=None):
    """(items * G, chunks) partial rows of dL/drgain[b][g] = <gx[b] [+ gxb[b]], tau[band(b) G + g]> for
    ``mlp_gains_bwd(ggains_parts=...)`` (tf_rows_sum of them = dL/drgain)."""
    _need_gpu(gx, tau)
    gx, tau = _f(gx), _f(tau)
    S = nbands * G
    if items % nbands or tuple(gx.shape) != (((items + 1) // 2, n, 2) if in_pairs else (items, n)) \
            or tuple(tau.shape) != (((S + 1) // 2, n, 2) if tau_pairs else (S, n)):
        raise RuntimeError("lin_gain_dots: signals do not match the stated layouts")
    if gxb is not None:
        gxb = _f(gxb)
        if gxb.shape != gx.shape:
            raise RuntimeError("lin_gain_dots: gxb must have the shape of gx")
    lib = _lib.load()
    nch = lib.gfdn_lin_gain_chunks(n)
    if out is None:
        parts = torch.empty((items * G, nch), dtype=_f32, device=gx.device)
    else:                                  # (a wider buffer whose first columns receive these partial sums)
        parts = out
        if parts.dtype != _f32 or not parts.is_contiguous() or parts.dim() != 2 or parts.shape[0] != items * G \
                or parts.shape[1] < nch:
            raise RuntimeError("lin_gain_dots: out must be (items * G, >= chunks) contiguous float32")
    _lib.check(lib.gfdn_lin_gain_dots(_p(gx), _p(gxb), n, int(in_pairs), _p(tau), n, int(tau_pairs), nbands,
                                      items // nbands, G, n, _p(parts), parts.shape[1], _stream()), "gfdn_lin_gain_dots")
    return parts


def lin_supported(B: int, G: int) -> bool:
    """Receivers per band x groups the time-domain output stage's sums take (gfdn_lin_gamma*: the band's gains sit in a
    256-entry LDS table)."""
    return 0 < G <= 4 and 0 < B and B * G <= 256


def spec_supported(B: int, G: int) -> bool:
    """... and the EDR loss on composed spectra with the receiver sums (gfdn_edr_lin_*, gfdn_lin_gamma_dots): whole receiver
    pairs, at most 64 receivers per band."""
    return lin_supported(B, G) and B % 2 == 0 and B <= 64


def lin_gamma_dots_tiles(n: int) -> int:
    return _lib.load().gfdn_lin_gamma_dots_tiles(int(n))


def lin_gamma_dots(g2, rgain, nbands: int, n: int, tau2, parts, win_start: int, win_len: int, base=None,
                   slot_of_time=None, band_win_len=None, base_b=None):
    """lin_gamma (pairs in, pairs out, ``base``, ``slot_of_time``) and lin_gain_dots in ONE sweep over gradient signals
    that are nonzero on [win_start, win_start + win_len) only (``band_win_len``: per-band lengths, int32 device): samples
    outside the window are not read.  ``parts`` (items * G, >= lin_gamma_dots_tiles(n)) receives the dot-product partial sums
    in its first columns.  Returns gamma."""
    _need_gpu(g2, rgain, tau2, parts)
    g2, rgain, tau2 = _f(g2), _f(rgain), _f(tau2)
    items, G = rgain.shape
    S = nbands * G
    lib = _lib.load()
    tiles = lib.gfdn_lin_gamma_dots_tiles(n)
    if items % nbands or tuple(g2.shape) != (items // 2, n, 2) or tuple(tau2.shape) != ((S + 1) // 2, n, 2) \
            or parts.dtype != _f32 or not parts.is_contiguous() or parts.shape[0] != items * G or parts.shape[1] < tiles:
        raise RuntimeError("lin_gamma_dots: shapes do not match")
    gamma = torch.empty(((S + 1) // 2, n, 2), dtype=_f32, device=g2.device)
    if S % 2:
        gamma[-1].zero_()
    for bb in (base, base_b):
        if bb is not None and (bb.dtype != _f32 or not bb.is_contiguous() or bb.shape != gamma.shape):
            raise RuntimeError("lin_gamma_dots: base / base_b must be shaped like gamma")
    _lib.check(lib.gfdn_lin_gamma_dots(_p(g2), n, _p(rgain), nbands, items // nbands, G, n, _p(tau2), n, _p(base), n,
                                       _p(slot_of_time), _p(gamma), n, _p(parts), parts.shape[1], int(win_start),
                                       int(win_len), _p(band_win_len), _p(base_b), _stream()), "gfdn_lin_gamma_dots")
    return gamma


def lin_gain_chunks(n: int) -> int:
    return _lib.load().gfdn_lin_gain_chunks(int(n))


# ---- the EDR loss on linearly composed short-time spectra (csrc/edrlin.hip) ------------------------------------------
def spec_tile(P: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """(..., nframes, nfreq) planes between the plain cell order (m nfreq + f) and the TILED one of csrc/edrlin.hip
    (frequency blocks of 256, a block's frames contiguous); the result has the same shape (a flat re-ordering of the last
    two axes).  One-off use: dataset stores and tests."""
    nframes, nfreq = P.shape[-2], P.shape[-1]
    lead = P.shape[:-2]
    flat = P.reshape(*lead, nframes * nfreq)
    out = torch.empty_like(flat)
    off = 0
    for f0 in range(0, nfreq, 256):
        w = min(256, nfreq - f0)
        if inverse:
            out.view(*lead, nframes, nfreq)[..., f0:f0 + w] = flat[..., off:off + nframes * w].reshape(*lead, nframes, w)
        else:
            out[..., off:off + nframes * w] = P[..., f0:f0 + w].reshape(*lead, nframes * w)
        off += nframes * w
    return out.view(*lead, nframes, nfreq)


def stft_pairs_spectrum(x2, items: int, win: int, tiled: bool = False) -> torch.Tensor:
    """x2 (ceil(items / 2), T, 2) pair-interleaved signals -> S (items, nframes, win / 2 + 1) complex64 (win = 4096);
    ``tiled``: the planes in the tiled cell order (spec_tile)."""
    _need_gpu(x2)
    if x2.dtype != _f32 or not x2.is_contiguous() or x2.dim() != 3 or x2.shape[2] != 2 or x2.shape[0] != (items + 1) // 2:
        raise RuntimeError("stft_pairs_spectrum: x2 must be contiguous float32 (ceil(items / 2), T, 2)")
    T = x2.shape[1]
    S = torch.empty((items, stft_nframes(T, win), win // 2 + 1), dtype=_c64, device=x2.device)
    _lib.check(_lib.load().gfdn_stft_pairs_spectrum(_p(x2), T, T, items, win, _p(S), int(tiled), _stream()),
               "gfdn_stft_pairs_spectrum")
    return S


def stft_pairs_spectrum_bwd(G, n: int, items: int, win: int, base=None, tiled: bool = False, split_parity: bool = False):
    """Adjoint of stft_pairs_spectrum: gradient spectra G (items, nframes, win / 2 + 1) complex64 -> gx2
    (ceil(items / 2), n, 2) [+ base, same layout].  G (nsplit, items, nframes, win / 2 + 1): partial sets, added in order
    where they are loaded (edr_lin_loss_gsum)."""
    _need_gpu(G)
    G = _c(G)
    nsplit = 1
    if G.dim() == 4:
        nsplit, G = G.shape[0], G.reshape(G.shape[0] * G.shape[1], G.shape[2], G.shape[3])
    if tuple(G.shape) != (nsplit * items, stft_nframes(n, win), win // 2 + 1):
        raise RuntimeError("stft_pairs_spectrum_bwd: G must be ([nsplit,] items, nframes, win / 2 + 1)")
    gx2 = torch.empty(((items + 1) // 2, n, 2), dtype=_f32, device=G.device)
    if base is not None and (base.dtype != _f32 or not base.is_contiguous() or base.shape != gx2.shape):
        raise RuntimeError("stft_pairs_spectrum_bwd: base must be shaped like the result")
    gx2b = torch.empty_like(gx2) if split_parity else None     # (split_parity: ONE launch, returns (gx2, gx2b): their sum)
    _lib.check(_lib.load().gfdn_stft_pairs_spectrum_bwd(_p(G), n, items, win, _p(base), _p(gx2), n, int(tiled), int(nsplit),
                                                        _p(gx2b), _stream()), "gfdn_stft_pairs_spectrum_bwd")
    return (gx2, gx2b) if split_parity else gx2


def edr_lin_parts(nfreq: int, fused: bool = False) -> int:
    """Partial-sum columns per item of edr_lin_loss (``fused`` False) / edr_lin_loss_gsum."""
    lib = _lib.load()
    return lib.gfdn_edr_lin_band_parts(int(nfreq)) if fused else lib.gfdn_edr_lin_parts(int(nfreq))


def edr_lin_loss(Sd, rows, Stau, rgain, nbands: int, T_db, sum_abs, gscale: float = 1.0, want_grad: bool = True,
                 dots=None, col0: int = 0, tiled: bool = False):
    """EDR loss of items whose short-time spectra are Sd[rows[b]] + sum_g rgain[b][g] Stau[band(b) G + g] -- never formed in
    memory.  Sd (R, nframes, nfreq) c64, T_db (R, nframes, nfreq) f32 and sum_abs (R) share the row indirection.
    Returns (part (items, edr_lin_parts) -- the deferred partial sums of edr_loss(defer=True) --, gP (items, nframes, nfreq)
    = gscale / sum_abs dloss/d|S|^2 or None).  ``dots`` (items * G, >= col0 + edr_lin_parts): receives the EDR part of
    dL/drgain as partial sums in columns [col0, col0 + edr_lin_parts)."""
    _need_gpu(Sd, Stau, rgain, T_db)
    Sd, Stau, rgain = _c(Sd), _c(Stau), _f(rgain)
    items, G = rgain.shape
    R, nframes, nfreq = Sd.shape
    if items % nbands or tuple(Stau.shape) != (nbands * G, nframes, nfreq) or tuple(T_db.shape) != (R, nframes, nfreq) \
            or T_db.dtype != _f32 or not T_db.is_contiguous() or sum_abs.numel() != R:
        raise RuntimeError("edr_lin_loss: Sd / T_db (R, nframes, nfreq), Stau (nbands * G, nframes, nfreq), sum_abs (R)")
    rows = _rows(rows, items, R)
    if rows is None and R != items:
        raise RuntimeError("edr_lin_loss: one row per item (or pass rows)")
    lib = _lib.load()
    fblk = lib.gfdn_edr_lin_parts(nfreq)
    part = torch.empty((items, fblk), dtype=_f32, device=Sd.device)
    gP = torch.empty((items, nframes, nfreq), dtype=_f32, device=Sd.device) if want_grad else None
    ld = 0
    if dots is not None:
        if dots.dtype != _f32 or not dots.is_contiguous() or dots.dim() != 2 or dots.shape[0] != items * G \
                or dots.shape[1] < col0 + fblk:
            raise RuntimeError("edr_lin_loss: dots must be (items * G, >= col0 + parts) contiguous float32")
        ld = dots.shape[1]
    end = kernel_timer.bracket('k_edr_lin_cols', items)          # (bench.py's roofline leg: events on the launch stream)
    _lib.check(lib.gfdn_edr_lin_loss(_p(Sd), _p(rows), _p(Stau), _p(rgain), nbands, items // nbands, G, _p(T_db),
                                     _p(_f(sum_abs)), nframes, nfreq, float(gscale), int(want_grad), _p(gP), _p(part),
                                     _p(dots if want_grad else None), ld, int(col0), int(tiled), _stream()),
               "gfdn_edr_lin_loss")
    if end is not None:
        end.record()
    return part, gP


def edr_lin_loss_gsum(Sd, rows, Stau, rgain, nbands: int, T_db, sum_abs, gscale: float = 1.0, dots=None, col0: int = 0,
                      tiled: bool = False, nsplit: int = 1):
    """edr_lin_loss(want_grad=True) and edr_lin_gsum as ONE launch (k_edr_lin_wave) -> (part (items, edr_lin_parts(fused)),
    Gsum (nsplit, nbands G, nframes, nfreq)): a thread owns cells of the band's plane and walks the band's receivers,
    dL/d|S|^2 is never written, Sd is read once.  The band's receivers are cut into ``nsplit`` runs, one partial plane set
    each: stft_pairs_spectrum_bwd adds them."""
    _need_gpu(Sd, Stau, rgain, T_db)
    Sd, Stau, rgain = _c(Sd), _c(Stau), _f(rgain)
    items, G = rgain.shape
    R, nframes, nfreq = Sd.shape
    if items % nbands or tuple(Stau.shape) != (nbands * G, nframes, nfreq) or tuple(T_db.shape) != (R, nframes, nfreq) \
            or T_db.dtype != _f32 or not T_db.is_contiguous() or sum_abs.numel() != R:
        raise RuntimeError("edr_lin_loss_gsum: Sd / T_db (R, nframes, nfreq), Stau (nbands * G, nframes, nfreq), sum_abs (R)")
    rows = _rows(rows, items, R)
    if rows is None and R != items:
        raise RuntimeError("edr_lin_loss_gsum: one row per item (or pass rows)")
    lib = _lib.load()
    fblk = lib.gfdn_edr_lin_band_parts(nfreq)
    part = torch.empty((items, fblk), dtype=_f32, device=Sd.device)
    Gs = torch.empty((nsplit, nbands * G, nframes, nfreq), dtype=_c64, device=Sd.device)
    ld = 0
    if dots is not None:
        if dots.dtype != _f32 or not dots.is_contiguous() or dots.dim() != 2 or dots.shape[0] != items * G \
                or dots.shape[1] < col0 + fblk:
            raise RuntimeError("edr_lin_loss_gsum: dots must be (items * G, >= col0 + parts) contiguous float32")
        ld = dots.shape[1]
    end = kernel_timer.bracket('k_edr_lin_wave', items)      # (bench.py's roofline leg)
    _lib.check(lib.gfdn_edr_lin_loss_gsum(_p(Sd), _p(rows), _p(Stau), _p(rgain), nbands, items // nbands, G, _p(T_db),
                                          _p(_f(sum_abs)), nframes, nfreq, float(gscale), _p(part), fblk, _p(dots), ld,
                                          int(col0), _p(Gs), int(nsplit), int(tiled), _stream()),
               "gfdn_edr_lin_loss_gsum")
    if end is not None:
        end.record()
    return part, Gs


def edr_lin_gsum(Sd, rows, Stau, rgain, nbands: int, gP) -> torch.Tensor:
    """Gsum (nbands * G, nframes, nfreq) complex64 = sum over the band's items of rgain[b][g] 2 gP[b] S[b] -- the gradient
    spectra whose adjoint STFT is the EDR part of dL/dtau."""
    _need_gpu(Sd, Stau, rgain, gP)
    Sd, Stau, rgain, gP = _c(Sd), _c(Stau), _f(rgain), _f(gP)
    items, G = rgain.shape
    R, nframes, nfreq = Sd.shape
    if items % nbands or tuple(Stau.shape) != (nbands * G, nframes, nfreq) or tuple(gP.shape) != (items, nframes, nfreq):
        raise RuntimeError("edr_lin_gsum: shapes do not match")
    rows = _rows(rows, items, R)
    Gs = torch.empty((nbands * G, nframes, nfreq), dtype=_c64, device=Sd.device)
    _lib.check(_lib.load().gfdn_edr_lin_gsum(_p(Sd), _p(rows), _p(Stau), _p(rgain), nbands, items // nbands, G, _p(gP),
                                             nframes, nfreq, _p(Gs), _stream()), "gfdn_edr_lin_gsum")
    return Gs


def edc_loss_pairs_lin(xd, rows, tau2, rgain, nbands: int, n: int, start: int, length: int, T_db, maskw=None,
                       inv_count: float = 1.0, gscale: float = 1.0, want_grad: bool = True, trows=None, item_len=None,
                       fill_outside: bool = True):
    """edc_loss_pairs on signals that are never stored (x[b] = xd[rows[b]] + sum_g rgain[b][g] tau[band G + g], formed where
    the scans read them) -> (loss_item (items,), g2 (ceil(items / 2), n, 2) or None).  ``fill_outside=False``: g2 is
    written on each item's window only (a consumer that reads nothing else: lin_gamma_dots)."""
    _need_gpu(xd, tau2, rgain, T_db)
    xd, tau2, rgain = _f(xd), _f(tau2), _f(rgain)
    items, G = rgain.shape
    S = nbands * G
    if items % nbands or xd.shape[1] < n or tuple(tau2.shape) != ((S + 1) // 2, n, 2):
        raise RuntimeError("edc_loss_pairs_lin: rgain (nbands * B, G), xd (R, >= n), tau2 (ceil(nbands G / 2), n, 2)")
    rows = _rows(rows, items, xd.shape[0])
    trows = _rows(trows, items, T_db.shape[0])
    if T_db.dtype != _f32 or not T_db.is_contiguous() or (item_len is None and T_db.shape[-1] != length) \
            or (trows is None and T_db.shape[0] != items):
        raise RuntimeError("edc_loss_pairs_lin: target shape does not match the window")
    maskw = None if maskw is None else _f(maskw)
    ld_mask = 0
    B = items // nbands
    if item_len is not None:
        _, ld_mask = _edc_bands(item_len, items, B, maskw, length, T_db, "edc_loss_pairs_lin")
    loss_item = torch.empty(items, dtype=_f32, device=xd.device)
    g2 = torch.empty(((items + 1) // 2, n, 2), dtype=_f32, device=xd.device) if want_grad else None
    xw = torch.empty(((items + 1) // 2, n, 2), dtype=_f32, device=xd.device)       # (window samples, written by the launch)
    lib = _lib.load()
    work = _work(lib.gfdn_edc_work_bytes(items + 1), xd.device)
    _lib.check(lib.gfdn_edc_loss_pairs_lin(_p(xd), xd.stride(0), _p(rows), _p(tau2), n, _p(rgain), nbands, B, G, n, start,
                                           length, _p(item_len), _p(T_db), T_db.shape[-1], _p(trows), _p(maskw), ld_mask,
                                           float(inv_count), float(gscale), _p(loss_item), _p(g2), int(fill_outside), _p(xw),
                                           _p(work), _stream()), "gfdn_edc_loss_pairs_lin")
    return loss_item, g2


def edc_lin_one_supported(length: int, G: int) -> bool:
    """Whether edc_lin_one takes a window of ``length`` samples and G groups per band."""
    return 0 < length <= _lib.load().gfdn_edc_lin_one_max_len() and 0 < G <= 4


def edc_lin_one(xd, rows, tau2, rgain, nbands: int, n: int, start: int, length: int, T_db, maskw=None,
                inv_count: float = 1.0, gscale: float = 1.0, want_grad: bool = True, trows=None, item_len=None,
                dots=None, col: int = 0):
    """The EDC term on x[b] = xd[rows[b]] + sum_g rgain[b][g] tau[band G + g] in ONE launch, one workgroup per item
    (csrc/edcone.hip) -> (loss_item (items,), gx (items, length) or None): gx[b][j] = dL/dx[b][start + j] on the item's
    window (columns behind a shorter per-item window are not written).  ``dots`` (items * G, cols) float32: column ``col``
    receives the EDC part of dL/drgain, <dL/dx[b], tau_g>."""
    _need_gpu(xd, tau2, rgain, T_db)
    xd, tau2, rgain = _f(xd), _f(tau2), _f(rgain)
    items, G = rgain.shape
    S = nbands * G
    if items % nbands or xd.shape[1] < n or tuple(tau2.shape) != ((S + 1) // 2, n, 2):
        raise RuntimeError("edc_lin_one: rgain (nbands * B, G), xd (R, >= n), tau2 (ceil(nbands G / 2), n, 2)")
    rows = _rows(rows, items, xd.shape[0])
    trows = _rows(trows, items, T_db.shape[0])
    if T_db.dtype != _f32 or not T_db.is_contiguous() or (item_len is None and T_db.shape[-1] != length) \
            or (trows is None and T_db.shape[0] != items) or start + length > n:
        raise RuntimeError("edc_lin_one: target shape does not match the window")
    maskw = None if maskw is None else _f(maskw)
    ld_mask = 0
    B = items // nbands
    if item_len is not None:
        _, ld_mask = _edc_bands(item_len, items, B, maskw, length, T_db, "edc_lin_one")
    elif maskw is not None and maskw.numel() < length:
        raise RuntimeError("edc_lin_one: mask weights shorter than the window")
    if dots is not None and (dots.dtype != _f32 or not dots.is_contiguous() or dots.dim() != 2
                             or dots.shape[0] != items * G or not 0 <= col < dots.shape[1]):
        raise RuntimeError("edc_lin_one: dots must be a contiguous float32 (items * G, cols) tensor, col inside it")
    loss_item = torch.empty(items, dtype=_f32, device=xd.device)
    gx = torch.empty((items, length), dtype=_f32, device=xd.device) if want_grad else None
    end = kernel_timer.bracket('k_edc_lin_one', items)       # (bench.py's roofline leg)
    _lib.check(_lib.load().gfdn_edc_lin_one(_p(xd), xd.stride(0), _p(rows), _p(tau2), n, _p(rgain), nbands, B, G, int(start),
                                            int(length), _p(item_len), _p(T_db), T_db.shape[-1], _p(trows), _p(maskw),
                                            ld_mask, float(inv_count), float(gscale), _p(loss_item), _p(gx), int(length),
                                            _p(dots if want_grad else None), 0 if dots is None else dots.shape[1], int(col),
                                            _stream()), "gfdn_edc_lin_one")
    if end is not None:
        end.record()
    return loss_item, gx


def lin_gamma_win(gx, rgain, nbands: int, n: int, win_start: int, win_len: int, base=None, base_b=None,
                  slot_of_time=None, band_win_len=None) -> torch.Tensor:
    """gamma (ceil(nbands G / 2), n, 2) = [base + base_b +] sum_b rgain[b][g] gx[b] with gx (items, >= win_len) the
    window-only rows of edc_lin_one (sample win_start + j at column j); ``slot_of_time``: written in the adjoint pair
    transform's slot order (irfft_odd_pairs_bwd(tslots=True))."""
    _need_gpu(gx, rgain)
    rgain = _f(rgain)
    items, G = rgain.shape
    S = nbands * G
    if gx.dtype != _f32 or not gx.is_contiguous() or gx.dim() != 2 or gx.shape[0] != items or gx.shape[1] < win_len \
            or items % nbands:
        raise RuntimeError("lin_gamma_win: gx must be a contiguous float32 (items, >= win_len) tensor")
    gamma = torch.empty(((S + 1) // 2, n, 2), dtype=_f32, device=gx.device)
    if S % 2:
        gamma[-1].zero_()
    for bb in (base, base_b):
        if bb is not None and (bb.dtype != _f32 or not bb.is_contiguous() or bb.shape != gamma.shape):
            raise RuntimeError("lin_gamma_win: base / base_b must be shaped like gamma")
    _lib.check(_lib.load().gfdn_lin_gamma_win(_p(gx), gx.shape[1], _p(rgain), nbands, items // nbands, G, n, int(win_start),
                                              int(win_len), _p(band_win_len), _p(base), _p(base_b), n, _p(slot_of_time),
                                              _p(gamma), n, _stream()), "gfdn_lin_gamma_win")
    return gamma


def lin_merge_slots(a2, b2=None, c2=None, slot_of_time=None) -> torch.Tensor:
    """a2 [+ b2 [+ c2]] -- pair-interleaved signal rows (rows, n, 2) -- written in the adjoint pair transform's slot order
    (``slot_of_time``; None: time order)."""
    _need_gpu(a2)
    for t in (a2, b2, c2):
        if t is not None and (t.dtype != _f32 or not t.is_contiguous() or t.shape != a2.shape or t.dim() != 3
                              or t.shape[2] != 2):
            raise RuntimeError("lin_merge_slots: contiguous float32 (rows, n, 2) tensors of one shape expected")
    out = torch.empty_like(a2)
    _lib.check(_lib.load().gfdn_lin_merge_slots(_p(a2), _p(b2), _p(c2), a2.shape[0], a2.shape[1], a2.shape[1],
                                                _p(slot_of_time), _p(out), a2.shape[1], _stream()), "gfdn_lin_merge_slots")
    return out


def tf_rows_sum(part: torch.Tensor) -> torch.Tensor:
    """part (..., cols) float32 -> sums over the last axis (one wavefront per row, fixed order)."""
    _need_gpu(part)
    part = _f(part)
    cols = part.shape[-1]
    out = torch.empty(part.shape[:-1], dtype=_f32, device=part.device)
    _lib.check(_lib.load().gfdn_tf_rows_sum(_p(part), cols, out.numel(), _p(out), _stream()), "gfdn_tf_rows_sum")
    return out


def irfft_odd_pairs_bwd(g2, n: int, batch: int, g2b=None, out=None, gains=None, tslots: bool = False, g2c=None):
    """Adjoint of irfft_odd_fwd(slots=True, pairs=True): g2 (ceil(batch / 2), n, 2) f32 pair-interleaved gradients
    [+ g2b, summed on load] -> gX (batch, (n + 1) / 2) c64 in slot order (``out``: where to write it).
    ``gains`` = (Tquad (nbands, (n+1)/2, 4), filt (nbands, (n+1)/2) or None, nbands, G): the gains pass of the output
    stage's adjoint rides the last pass -- returns (gX, gpart (batch, G, parts)); tf_rows_sum(gpart) = dL/drgain."""
    _need_gpu(g2)
    npairs = (batch + 1) // 2
    for t in (g2, g2b, g2c):
        if t is not None and (t.dtype != _f32 or not t.is_contiguous() or tuple(t.shape) != (npairs, n, 2)):
            raise RuntimeError("irfft_odd_pairs_bwd: pair-interleaved float32 gradients (pairs, n, 2) expected")
    lib = _lib.load()
    table = bluestein_table(n, g2.device)
    ldx = (n + 1) // 2
    if out is not None and (out.dtype != _c64 or not out.is_contiguous() or tuple(out.shape) != (batch, ldx)):
        raise RuntimeError("irfft_odd_pairs_bwd: out must be a contiguous complex64 (batch, (n + 1) / 2) tensor")
    gX = torch.empty((batch, ldx), dtype=_c64, device=g2.device) if out is None else out
    work = _work(lib.gfdn_bluestein_work_bytes(n, batch), g2.device)
    if tslots:                             # (g2 in the transform's own order: lin_gamma(slot_of_time=...))
        if gains is not None or (g2c is not None and g2b is None):
            raise RuntimeError("irfft_odd_pairs_bwd(tslots=True) takes up to three gradient signals (g2, g2b, g2c)")
        if g2b is not None:                # (parts of one gradient signal, all slot-ordered: summed where they are loaded)
            _lib.check(lib.gfdn_irfft_odd_pairs_bwd_tslots3(_p(table), n, _p(g2), _p(g2b), _p(g2c), n, batch, _p(gX), ldx,
                                                            _p(work), _stream()), "gfdn_irfft_odd_pairs_bwd_tslots3")
            return gX
        _lib.check(lib.gfdn_irfft_odd_pairs_bwd_tslots(_p(table), n, _p(g2), n, batch, _p(gX), ldx, _p(work), _stream()),
                   "gfdn_irfft_odd_pairs_bwd_tslots")
        return gX
    if g2c is not None:
        raise RuntimeError("irfft_odd_pairs_bwd: g2c goes with tslots=True")
    if gains is not None:
        Tq, filt, nbands, G = gains
        Tq = _c(Tq)
        filt = None if filt is None else _c(filt)
        if tuple(Tq.shape) != (nbands, ldx, 4) or batch % nbands:
            raise RuntimeError("irfft_odd_pairs_bwd: gains = (Tquad (nbands, (n+1)/2, 4), filt, nbands, G)")
        gpart = torch.empty((batch, G, lib.gfdn_irfft_odd_pairs_gains_parts(n)), dtype=_f32, device=g2.device)
        args = (_p(table), n, _p(g2), _p(g2b), n, batch, _p(gX), ldx, _p(Tq), ldx, _p(filt), ldx, nbands, G, _p(gpart),
                _p(work))
        if kernel_timer.active and kernel_timer.watch in _BLU_STAGES:
            for name, stage in _BLU_STAGES.items():
                end = kernel_timer.bracket(name, batch)
                _lib.check(lib.gfdn_irfft_odd_pairs_gains_bwd(*args, stage, _stream()),
                           "gfdn_irfft_odd_pairs_gains_bwd[%s]" % name)
                if end is not None:
                    end.record()
        else:
            _lib.check(lib.gfdn_irfft_odd_pairs_gains_bwd(*args, 7, _stream()), "gfdn_irfft_odd_pairs_gains_bwd")
        return gX, gpart
    if kernel_timer.active and kernel_timer.watch in _BLU_STAGES:
        _staged_bluestein(lib, table, n, g2, g2b, n, batch, gX, ldx, work, 1, 2)
        return gX
    _lib.check(lib.gfdn_irfft_odd_pairs_bwd(_p(table), n, _p(g2), _p(g2b), n, batch, _p(gX), ldx, _p(work),
                                            _stream()), "gfdn_irfft_odd_pairs_bwd")
    return gX


def irfft_odd_bwd(gx, n: int, ldx: int, gx2=None, slots: bool = False) -> torch.Tensor:
    """gx (batch, n) f32 -> gX (batch, ldx) c64 (adjoint of irfft_odd_fwd).  ``gx2``: optional second
    gradient of the same shape; the transform is applied to gx + gx2 (summed on load).
    ``slots``: gX comes back in slot order (ldx must be (n + 1) / 2)."""
    _need_gpu(gx)
    gx = _f(gx)
    batch = gx.shape[0]
    if gx2 is not None:
        _need_gpu(gx2)
        gx2 = _f(gx2)
        if gx2.shape != gx.shape:
            raise RuntimeError("irfft_odd_bwd: gx2 must have the shape of gx")
    lib = _lib.load()
    table = bluestein_table(n, gx.device)
    gX = torch.empty((batch, ldx), dtype=_c64, device=gx.device)
    work = _work(lib.gfdn_bluestein_work_bytes(n, batch), gx.device)
    if slots and ldx != (n + 1) // 2:
        raise RuntimeError("irfft_odd_bwd(slots=True): the slot-ordered gradient has (n + 1) / 2 columns")
    if kernel_timer.active and kernel_timer.watch in _BLU_STAGES:
        _staged_bluestein(lib, table, n, gx, gx2, gx.shape[1], batch, gX, ldx, work, 1, slots)
        return gX
    fn = lib.gfdn_irfft_odd_slots_bwd if slots else lib.gfdn_irfft_odd_bwd
    _lib.check(fn(_p(table), n, _p(gx), _p(gx2), gx.shape[1], batch, _p(gX), ldx, _p(work), _stream()),
               "gfdn_irfft_odd_bwd")
    return gX


# stage bit of gfdn_irfft_odd_stages by the kernel that runs it at the north-star length (n = 65 537: Rader on the
# 128 x 512 geometry; other lengths run k_blu_col_fwd / k_blu_row / k_blu_col_inv under the same bits)
_BLU_STAGES = {'k_blu_col128_fwd': 1, 'k_blu_row512': 2, 'k_blu_col128_inv': 4}


def _staged_bluestein(lib, table, n, src, src2, ld_in, batch, dst, ld_out, work, adjoint, slots=False):
    """Same three launches as the fused entry point, with HIP events around the watched one."""
    args = (_p(table), n, _p(src), _p(src2), ld_in, batch, _p(dst), ld_out, _p(work), adjoint)
    for name, stage in _BLU_STAGES.items():
        end = kernel_timer.bracket(name, batch)
        _lib.check(lib.gfdn_irfft_odd_stages(*args, stage, int(slots), _stream()),
                   "gfdn_irfft_odd_stages[%s]" % name)
        if end is not None:
            end.record()


def irfft_pow2_fwd(X, n: int) -> torch.Tensor:
    _need_gpu(X)
    X = _c(X)
    batch, ldx = X.shape
    lib = _lib.load()
    x = torch.empty((batch, n), dtype=_f32, device=X.device)
    work = _work(lib.gfdn_irfft_pow2_work_bytes(n, batch), X.device)
    end = kernel_timer.bracket('irfft_pow2_fwd', batch)        # (both passes of the transform as one unit)
    _lib.check(lib.gfdn_irfft_pow2_fwd(n, _p(X), ldx, batch, _p(x), n, _p(work), _stream()),
               "gfdn_irfft_pow2_fwd")
    if end is not None:
        end.record()
    return x


def irfft_pow2_bwd(gx, n: int, window=None) -> torch.Tensor:
    """gX = d<gx, irfft(X, n)>/dX.  ``window`` = (lo, hi): gx vanishes outside the samples [lo, hi) and whatever the
    buffer holds there is not read (n = 131 072)."""
    _need_gpu(gx)
    gx = _f(gx)
    batch = gx.shape[0]
    lib = _lib.load()
    gX = torch.empty((batch, n // 2 + 1), dtype=_c64, device=gx.device)
    work = _work(lib.gfdn_irfft_pow2_work_bytes(n, batch), gx.device)
    if window is not None:
        _lib.check(lib.gfdn_irfft_pow2_bwd_window(n, _p(gx), gx.shape[1], batch, int(window[0]), int(window[1]), _p(gX),
                                                  n // 2 + 1, _p(work), _stream()), "gfdn_irfft_pow2_bwd_window")
        return gX
    _lib.check(lib.gfdn_irfft_pow2_bwd(n, _p(gx), gx.shape[1], batch, _p(gX), n // 2 + 1, _p(work),
                                       _stream()), "gfdn_irfft_pow2_bwd")
    return gX


def rfft_pow2(x, n: int) -> torch.Tensor:
    """x (batch, T <= n) float32 -> rfft(x, n) (batch, n/2+1) complex64."""
    _need_gpu(x)
    x = _f(x)
    batch, T = x.shape
    lib = _lib.load()
    X = torch.empty((batch, n // 2 + 1), dtype=_c64, device=x.device)
    work = _work(lib.gfdn_irfft_pow2_work_bytes(n, batch), x.device)
    _lib.check(lib.gfdn_rfft_pow2(n, _p(x), T, T, batch, _p(X), n // 2 + 1, _p(work), _stream()),
               "gfdn_rfft_pow2")
    return X


def rfft_pow2_f64(x, n: int, kout: Optional[int] = None) -> torch.Tensor:
    """x (batch, T <= n) float64 -> the first ``kout`` (default n / 2 + 1) bins of rfft(x, n) as complex128 (csrc/fft64.hip:
    float64 radix-2 passes -- dataset constants only, speed is not the point)."""
    _need_gpu(x)
    if x.dtype != torch.float64 or x.dim() != 2 or n < 2 or n & (n - 1):
        raise RuntimeError("rfft_pow2_f64: a (batch, T) float64 tensor and a power-of-two length expected")
    x = x.contiguous()
    batch, T = x.shape
    kout = n // 2 + 1 if kout is None else int(kout)
    lib = _lib.load()
    X = torch.empty((batch, kout), dtype=torch.complex128, device=x.device)
    work = _work(lib.gfdn_f64_fft_work_bytes(batch, n), x.device)
    _lib.check(lib.gfdn_rfft_pow2_f64(_p(x), T, T, batch, n, _p(X), kout, _p(work), _stream()), "gfdn_rfft_pow2_f64")
    return X


_IRFFT64_PLANS = {}


def irfft_odd_f64(X, n: int, filt=None, out_dtype=torch.float32, chunk: int = 32) -> torch.Tensor:
    """X (batch, >= (n + 1) / 2) complex128 [times ``filt`` (>= (n + 1) / 2) complex128] -> irfft(., n) for an odd n, evaluated
    in float64 (Bluestein on radix-2 double passes, csrc/fft64.hip) and rounded ONCE to ``out_dtype`` (float32 / float64).
    For stores that are built once per dataset (BandStackedDataset.direct_time): the reference's irfft runs on complex128
    spectra (losses.py:207-213, :442-445)."""
    _need_gpu(X)
    if X.dtype != torch.complex128 or X.dim() != 2 or n < 3 or not (n & 1) or X.shape[1] < (n + 1) // 2:
        raise RuntimeError("irfft_odd_f64: a (batch, >= (n + 1) / 2) complex128 tensor and an odd length expected")
    if filt is not None and (filt.dtype != torch.complex128 or filt.numel() < (n + 1) // 2 or not filt.is_contiguous()):
        raise RuntimeError("irfft_odd_f64: filt must be a contiguous complex128 vector with at least (n + 1) / 2 entries")
    if out_dtype not in (torch.float32, torch.float64):
        raise RuntimeError("irfft_odd_f64: float32 or float64 output")
    X = X.contiguous()
    lib = _lib.load()
    M = lib.gfdn_irfft_odd_f64_length(n)
    key = (n, str(X.device))
    if key not in _IRFFT64_PLANS:
        bhat = torch.empty(M, dtype=torch.complex128, device=X.device)
        work = _work(lib.gfdn_f64_fft_work_bytes(1, M), X.device)
        _lib.check(lib.gfdn_irfft_odd_f64_plan(n, _p(bhat), _p(work), _stream()), "gfdn_irfft_odd_f64_plan")
        _IRFFT64_PLANS[key] = bhat
    bhat = _IRFFT64_PLANS[key]
    batch = X.shape[0]
    out = torch.empty((batch, n), dtype=out_dtype, device=X.device)
    work = _work(lib.gfdn_f64_fft_work_bytes(min(chunk, batch), M), X.device)
    for r0 in range(0, batch, chunk):
        r1 = min(r0 + chunk, batch)
        o = out[r0:r1]
        _lib.check(lib.gfdn_irfft_odd_f64(_p(X[r0:r1]), X.stride(0), _p(filt), r1 - r0, n, _p(bhat),
                                          _p(o) if out_dtype == torch.float32 else None,
                                          _p(o) if out_dtype == torch.float64 else None, n, _p(work), _stream()),
                   "gfdn_irfft_odd_f64")
    return out


def sh_to_directional(A, H, adjoint: bool = False) -> torch.Tensor:
    """A (J,C) real, H (B,C,K) complex -> (B,J,K); adjoint maps (B,J,K) -> (B,C,K)."""
    _need_gpu(A, H)
    A, H = _f(A), _c(H)
    J, C = A.shape
    B, nin, K = H.shape
    assert nin == (J if adjoint else C)
    out = torch.empty((B, C if adjoint else J, K), dtype=_c64, device=H.device)
    _lib.check(_lib.load().gfdn_sh_to_directional(_p(A), J, C, K, B, _p(H), _p(out), int(adjoint),
                                                  _stream()), "gfdn_sh_to_directional")
    return out


# ------------------------------------------------------------------------------------------------
def stft_nframes(T: int, win: int) -> int:
    nf = _lib.load().gfdn_stft_nframes(int(T), int(win))
    if nf <= 0:
        raise RuntimeError(f"stft: signal of {T} samples too short for window {win} (or win not 2^p)")
    return nf


def stft_power(x, win: int, zero_buf=None) -> torch.Tensor:
    """x (batch, T) f32 -> P (batch, nframes, win/2+1) = |STFT|^2.  ``zero_buf``: a (batch, T) float32
    buffer cleared by the same launch (the accumulation buffer of stft_power_bwd)."""
    _need_gpu(x)
    x = _f(x)
    batch, T = x.shape
    nf = stft_nframes(T, win)
    if zero_buf is not None:
        _need_gpu(zero_buf)
        if zero_buf.dtype != _f32 or zero_buf.shape != x.shape or not zero_buf.is_contiguous():
            raise RuntimeError("stft_power: zero_buf must be a contiguous float32 buffer shaped like x")
    P = torch.empty((batch, nf, win // 2 + 1), dtype=_f32, device=x.device)
    _lib.check(_lib.load().gfdn_stft_power(_p(x), T, T, batch, win, _p(P), _p(zero_buf), _stream()),
               "gfdn_stft_power")
    return P


def stft_power_bwd(x, win: int, gP, gx_accum: torch.Tensor) -> torch.Tensor:
    """Accumulates d<gP, P>/dx into gx_accum (batch, T) and returns it."""
    _need_gpu(x, gP, gx_accum)
    x = _f(x)
    batch, T = x.shape
    assert gx_accum.dtype == _f32 and gx_accum.is_contiguous() and gx_accum.shape == x.shape
    _lib.check(_lib.load().gfdn_stft_power_bwd(_p(x), T, T, batch, win, _p(gP), _p(gx_accum),
                                               _stream()), "gfdn_stft_power_bwd")
    return gx_accum


def stft_power_pairs(x2, items: int, win: int, zero_buf=None) -> torch.Tensor:
    """x2 (ceil(items / 2), T, 2) pair-interleaved signals -> P (items, nframes, win/2+1) = |STFT|^2 (win = 4096).
    ``zero_buf``: a buffer shaped like x2, cleared by the same launch (accumulation buffer of the adjoint)."""
    _need_gpu(x2)
    if x2.dtype != _f32 or not x2.is_contiguous() or x2.dim() != 3 or x2.shape[2] != 2 or x2.shape[0] != (items + 1) // 2:
        raise RuntimeError("stft_power_pairs: x2 must be contiguous float32 (ceil(items / 2), T, 2)")
    T = x2.shape[1]
    nf = stft_nframes(T, win)
    if zero_buf is not None and (zero_buf.dtype != _f32 or zero_buf.shape != x2.shape or not zero_buf.is_contiguous()):
        raise RuntimeError("stft_power_pairs: zero_buf must be shaped like x2")
    P = torch.empty((items, nf, win // 2 + 1), dtype=_f32, device=x2.device)
    _lib.check(_lib.load().gfdn_stft_power_pairs(_p(x2), T, T, items, win, _p(P), _p(zero_buf), _stream()),
               "gfdn_stft_power_pairs")
    return P


def stft_power_pairs_bwd(x2, items: int, win: int, gP, base=None, out=None, phase=None) -> torch.Tensor:
    """-> base + d<gP, P>/dx2, shaped like x2 (``base``: another gradient of the same layout, e.g. the EDC loss's;
    ``out`` may be ``base`` itself).  Stored, not accumulated: ``out`` needs no clearing.
    ``phase`` 0 / 1: one of the two launches (even / odd frames) with the base added by the SECOND one, so that
    the first does not wait for the producer of ``base`` (``out`` must then not alias ``base``)."""
    _need_gpu(x2, gP)
    out = torch.empty_like(x2) if out is None else out
    if phase is not None:
        if phase == 1 and base is not None and base.data_ptr() == out.data_ptr():
            raise RuntimeError("stft_power_pairs_bwd(phase=1): out must not alias base")
        for t in (base, out):
            if t is not None and (t.dtype != _f32 or not t.is_contiguous() or t.shape != x2.shape or not t.is_cuda):
                raise RuntimeError("stft_power_pairs_bwd: base / out must be shaped like x2")
        _lib.check(_lib.load().gfdn_stft_power_pairs_bwd_phase(_p(x2), x2.shape[1], x2.shape[1], items, win, _p(gP),
                                                               _p(base), _p(out), int(phase), _stream()),
                   "gfdn_stft_power_pairs_bwd_phase")
        return out
    for t in (base, out):
        if t is not None and (t.dtype != _f32 or not t.is_contiguous() or t.shape != x2.shape or not t.is_cuda):
            raise RuntimeError("stft_power_pairs_bwd: base / out must be shaped like x2")
    T = x2.shape[1]
    _lib.check(_lib.load().gfdn_stft_power_pairs_bwd(_p(x2), T, T, items, win, _p(gP), _p(base), _p(out),
                                                     _stream()), "gfdn_stft_power_pairs_bwd")
    return out


def edc_loss_model(x, start: int, length: int, amps, env, maskw=None, inv_count: float = 1.0, gscale: float = 1.0,
                   want_grad: bool = True):
    """edc_loss against the common-slope model: target EDC of item b = sum_k amps[b][k] env[k][:length] (in dB inside
    the kernel).  x (B, T) f32, amps (B, S) f32, env (S, >= length) f32 -> (loss_item (B,), gx like x or None)."""
    _need_gpu(x, amps, env)
    x, amps, env = _f(x), _f(amps), _f(env)
    B, T = x.shape
    S = amps.shape[1]
    if amps.shape[0] != B or env.shape[0] != S or env.shape[1] < length:
        raise RuntimeError("edc_loss_model: amps (B, S), env (S, >= length)")
    maskw = None if maskw is None else _f(maskw)
    loss_item = torch.empty(B, dtype=_f32, device=x.device)
    gx = torch.empty_like(x) if want_grad else None
    lib = _lib.load()
    work = _work(lib.gfdn_edc_work_bytes(B), x.device)
    _lib.check(lib.gfdn_edc_loss_model(_p(x), T, B, start, length, _p(amps), S, _p(env), env.shape[1], _p(maskw),
                                       float(inv_count), float(gscale), _p(loss_item), _p(gx), _p(work), _stream()),
               "gfdn_edc_loss_model")
    return loss_item, gx


def rownorm_fwd(w, eps: float = 1e-6) -> torch.Tensor:
    """w (..., len) f32 -> w / (||w||_2 + eps) over the last axis."""
    _need_gpu(w)
    w = _f(w)
    y = torch.empty_like(w)
    _lib.check(_lib.load().gfdn_rownorm_fwd(_p(w), w.numel() // w.shape[-1], w.shape[-1], float(eps), _p(y), _stream()),
               "gfdn_rownorm_fwd")
    return y


def rownorm_bwd(w, gy, eps: float = 1e-6) -> torch.Tensor:
    _need_gpu(w, gy)
    w, gy = _f(w), _f(gy)
    gw = torch.empty_like(w)
    _lib.check(_lib.load().gfdn_rownorm_bwd(_p(w), w.numel() // w.shape[-1], w.shape[-1], float(eps), _p(gy), _p(gw),
                                            _stream()), "gfdn_rownorm_bwd")
    return gw


def edc_mixed_supported(C: int, J: int, S: int = 1) -> bool:
    """Channel / direction / slope counts gfdn_edc_loss_model_mixed is built for."""
    return C in (1, 4, 9, 16) and J <= 16 and S <= 8


def edc_loss_model_mixed(x_sh, A, start: int, length: int, amps, env, maskw=None, inv_count: float = 1.0,
                         gscale: float = 1.0, want_grad: bool = True):
    """edc_loss_model on the directional signals A x_sh without forming them: x_sh (B, C, T) f32, A (J, C), amps (B J, S),
    env (S, >= length) -> (loss_item (B J,) TIMES gscale, gx_sh like x_sh or None).  gx_sh is written on the window samples
    [start, start + length) ONLY: hand it to irfft_pow2_bwd(..., window=(start, start + length))."""
    _need_gpu(x_sh, A, amps, env)
    x_sh, A, amps, env = _f(x_sh), _f(A), _f(amps), _f(env)
    B, C, T = x_sh.shape
    J, S = A.shape[0], amps.shape[1]
    if A.shape[1] != C or amps.shape[0] != B * J or env.shape[0] != S or env.shape[1] < length:
        raise RuntimeError("edc_loss_model_mixed: A (J, C), amps (B J, S), env (S, >= length)")
    maskw = None if maskw is None else _f(maskw)
    loss_item = torch.empty(B * J, dtype=_f32, device=x_sh.device)
    gx = torch.empty_like(x_sh) if want_grad else None
    lib = _lib.load()
    work = _work(lib.gfdn_edc_mixed_work_bytes(B, J, length), x_sh.device)
    args = (_p(x_sh), T, B, C, _p(A), J, start, length, _p(amps), S, _p(env), env.shape[1], _p(maskw), float(inv_count),
            float(gscale), _p(loss_item), _p(gx), _p(work))
    if want_grad and kernel_timer.active and kernel_timer.watch == 'k_em_bwd':     # events around the one kernel
        _lib.check(lib.gfdn_edc_loss_model_mixed_stages(*args, 1, _stream()), "gfdn_edc_loss_model_mixed_stages[1]")
        end = kernel_timer.bracket('k_em_bwd', B)
        _lib.check(lib.gfdn_edc_loss_model_mixed_stages(*args, 2, _stream()), "gfdn_edc_loss_model_mixed_stages[2]")
        if end is not None:
            end.record()
        return loss_item, gx
    _lib.check(lib.gfdn_edc_loss_model_mixed(*args, _stream()), "gfdn_edc_loss_model_mixed")
    return loss_item, gx


def group_sums_supported(G: int, nper: int) -> bool:
    return G * nper <= 64


def group_sums_fwd(Y, c, G: int, nper: int) -> torch.Tensor:
    """S (G, K) complex64 = sum over the group's lines of c_n Y[k][n]  (model.py:243-250)."""
    _need_gpu(Y, c)
    Y, c = _c(Y), _f(c)
    K, N = Y.shape
    if N != G * nper or c.numel() != N:
        raise RuntimeError("group_sums_fwd: Y (K, G nper), c (G nper,)")
    S = torch.empty((G, K), dtype=_c64, device=Y.device)
    _lib.check(_lib.load().gfdn_group_sums_fwd(_p(Y), K, G, nper, _p(c), _p(S), _stream()), "gfdn_group_sums_fwd")
    return S


def group_sums_bwd(Y, c, G: int, nper: int, gS):
    """(gY (K, N), gc (N,)) from gS (G, K)."""
    _need_gpu(Y, c, gS)
    Y, c, gS = _c(Y), _f(c), _c(gS)
    K, N = Y.shape
    lib = _lib.load()
    gY = torch.empty_like(Y)
    gc = torch.empty(N, dtype=_f32, device=Y.device)
    part = torch.empty(N * lib.gfdn_dirlin_line_tiles(K), dtype=_f32, device=Y.device)
    _lib.check(lib.gfdn_group_sums_bwd(_p(Y), K, G, nper, _p(c), _p(gS), _p(gY), _p(gc), _p(part), _stream()),
               "gfdn_group_sums_bwd")
    return gY, gc


# ---- the directional output stage in the time domain (csrc/dirlin.hip) -----------------------------------------------
def dirlin_supported(G: int, nper: int) -> bool:
    """Group / line counts the gfdn_dirlin_* kernels are built for."""
    return 1 <= G <= 4 and G * nper <= 64


def dirlin_lines_fwd(Y, c, filt=None) -> torch.Tensor:
    """Z (N, K) complex64 = c_n filt_k Y[k][n]: the delay-line responses Y (K, N) of the transposed solve as the N rows the
    inverse transform takes (model.py:1056-1088 without the receivers' weights)."""
    _need_gpu(Y, c, filt)
    Y, c = _c(Y), _f(c)
    filt = None if filt is None else _c(filt)
    K, N = Y.shape
    Z = torch.empty((N, K), dtype=_c64, device=Y.device)
    _lib.check(_lib.load().gfdn_dirlin_lines_fwd(_p(Y), K, N, _p(c), _p(filt), _p(Z), K, _stream()),
               "gfdn_dirlin_lines_fwd")
    return Z


def dirlin_lines_bwd(Y, c, filt, gZ):
    """(gY (K, N) complex64, gc (N,)) from gZ (N, K) = dL/dZ of dirlin_lines_fwd."""
    _need_gpu(Y, c, filt, gZ)
    Y, c, gZ = _c(Y), _f(c), _c(gZ)
    filt = None if filt is None else _c(filt)
    K, N = Y.shape
    lib = _lib.load()
    gY = torch.empty_like(Y)
    gc = torch.empty(N, dtype=_f32, device=Y.device)
    part = torch.empty(N * lib.gfdn_dirlin_line_tiles(K), dtype=_f32, device=Y.device)
    _lib.check(lib.gfdn_dirlin_lines_bwd(_p(Y), K, N, _p(c), _p(filt), _p(gZ), gZ.shape[1], _p(gY), _p(gc), _p(part),
                                         _stream()), "gfdn_dirlin_lines_bwd")
    return gY, gc


def dirlin_combine(tau, start: int, length: int, w, G: int, nper: int) -> torch.Tensor:
    """x (B, nper, ceil4(length)) f32: the receivers' SH-domain signals on the window [start, start + length) from the
    N = G nper line signals tau (N, n) and the SH weights w (B, N): x[b][l] = sum_g w[b][g nper + l] tau[g nper + l]."""
    _need_gpu(tau, w)
    tau, w = _f(tau), _f(w)
    B = w.shape[0]
    if tau.shape[0] != G * nper or w.shape[1] != G * nper:
        raise RuntimeError("dirlin_combine: tau (G nper, n), w (B, G nper)")
    Lp = (length + 3) & ~3
    x = torch.empty((B, nper, Lp), dtype=_f32, device=tau.device)
    _lib.check(_lib.load().gfdn_dirlin_combine(_p(tau), tau.shape[1], start, length, _p(w), B, G, nper, _p(x), Lp,
                                               _stream()), "gfdn_dirlin_combine")
    return x


def dirlin_gamma_dots(gx, length: int, tau, start: int, w, G: int, nper: int):
    """Adjoint of dirlin_combine: (gtau (N, n) -- WRITTEN ON THE WINDOW ONLY, for irfft_pow2_bwd(..., window=) --, gw (B, N))
    from gx (B, nper, >= length)."""
    _need_gpu(gx, tau, w)
    gx, tau, w = _f(gx), _f(tau), _f(w)
    B, N = w.shape
    lib = _lib.load()
    gtau = torch.empty_like(tau)
    gw = torch.empty((B, N), dtype=_f32, device=tau.device)
    part = torch.empty(B * N * lib.gfdn_dirlin_tiles(length), dtype=_f32, device=tau.device)
    _lib.check(lib.gfdn_dirlin_gamma_dots(_p(gx), gx.shape[-1], length, _p(tau), tau.shape[1], start, _p(w), B, G, nper,
                                          _p(gtau), tau.shape[1], _p(gw), _p(part), _stream()), "gfdn_dirlin_gamma_dots")
    return gtau, gw


def _edc_bands(item_len, items: int, items_per_band, maskw, length: int, T_db, what: str):
    """Arguments of the banded EDC entry points: (items_per_band, ld_mask) -- see gfdn_edc_loss_banded.
    ``maskw`` (bands, >= length) selects per-band mask rows, a vector is shared by all bands."""
    if item_len.dtype != torch.int32 or not item_len.is_contiguous() or item_len.numel() != items:
        raise RuntimeError(f"{what}: item_len must be a contiguous int32 tensor with one window length per item")
    ipb = items if items_per_band is None else int(items_per_band)
    if ipb <= 0 or items % ipb:
        raise RuntimeError(f"{what}: items_per_band must divide the number of items")
    ld_mask = 0
    if maskw is not None and maskw.ndim == 2:
        if maskw.shape[0] != items // ipb or maskw.shape[1] < length:
            raise RuntimeError(f"{what}: per-band mask weights must be (bands, >= window)")
        ld_mask = maskw.shape[1]
    elif maskw is not None and maskw.numel() < length:
        raise RuntimeError(f"{what}: mask weights shorter than the window")
    if T_db.shape[-1] < length:
        raise RuntimeError(f"{what}: target rows shorter than the longest window")
    return ipb, ld_mask


def edc_loss_pairs(x2, items: int, start: int, length: int, T_db, maskw=None, inv_count: float = 1.0,
                   gscale: float = 1.0, want_grad: bool = True, rows=None, out=None, item_len=None,
                   items_per_band=None):
    """edc_loss on pair-interleaved signals x2 (ceil(items / 2), T, 2) -> loss_item (items,), g2 like x2 or None.
    ``item_len`` (items,) int32: per-item window lengths <= ``length`` (band banks whose bands differ in T60max,
    gfdn_edc_loss_pairs_banded): T_db rows are then padded to >= length, ``maskw`` may be (bands, >= length)."""
    _need_gpu(x2, T_db)
    T = x2.shape[1]
    rows = _rows(rows, items, T_db.shape[0])
    if T_db.dtype != _f32 or not T_db.is_contiguous() or (item_len is None and T_db.shape[-1] != length) \
            or (rows is None and T_db.shape[0] != items):
        raise RuntimeError("edc_loss_pairs: target shape does not match the window")
    maskw = None if maskw is None else _f(maskw)
    if out is not None and (out.dtype != _f32 or not out.is_contiguous() or out.numel() != items):
        raise RuntimeError("edc_loss_pairs: out must be a contiguous float32 vector with one entry per item")
    loss_item = torch.empty(items, dtype=_f32, device=x2.device) if out is None else out
    g2 = torch.empty_like(x2) if want_grad else None      # (the kernel writes zeros for a missing partner)
    lib = _lib.load()
    work = _work(lib.gfdn_edc_work_bytes(items + 1), x2.device)
    if item_len is not None:
        ipb, ld_mask = _edc_bands(item_len, items, items_per_band, maskw, length, T_db, "edc_loss_pairs")
        _lib.check(lib.gfdn_edc_loss_pairs_banded(_p(x2), T, items, start, length, _p(item_len), _p(T_db),
                                                  T_db.shape[-1], _p(rows), _p(maskw), ld_mask, ipb, float(inv_count),
                                                  float(gscale), _p(loss_item), _p(g2), _p(work), _stream()),
                   "gfdn_edc_loss_pairs_banded")
        return loss_item, g2
    _lib.check(lib.gfdn_edc_loss_pairs(_p(x2), T, items, start, length, _p(T_db), _p(rows), _p(maskw),
                                       float(inv_count), float(gscale), _p(loss_item), _p(g2), _p(work),
                                       _stream()), "gfdn_edc_loss_pairs")
    return loss_item, g2


def edr_partial_cols(batch: int, nfreq: int) -> int:
    """Columns of the deferred partial sums of edr_loss(defer=True)."""
    return _lib.load().gfdn_edr_work_bytes(int(batch), int(nfreq)) // (4 * int(batch))


def edr_target(P: torch.Tensor):
    """In place: P (batch, nframes, nfreq) -> EDR dB.  Returns (T_db (= P), sum_abs (batch,))."""
    _need_gpu(P)
    assert P.dtype == _f32 and P.is_contiguous()
    batch, nframes, nfreq = P.shape
    lib = _lib.load()
    sum_abs = torch.empty(batch, dtype=_f32, device=P.device)
    work = _work(lib.gfdn_edr_work_bytes(batch, nfreq), P.device)
    _lib.check(lib.gfdn_edr_target(_p(P), batch, nframes, nfreq, _p(sum_abs), _p(work), _stream()),
               "gfdn_edr_target")
    return P, sum_abs


def edr_loss(P, T_db, sum_abs, wf=None, gscale: float = 1.0, want_grad: bool = True, rows=None,
             defer: bool = False, out=None):
    """In place on P (achieved |STFT|^2): returns loss_item (batch,); P becomes dloss/dP.
    ``rows``: item b compares against row rows[b] of the (all-receiver) target store.
    ``defer``: return the (batch, tiles) partial sums instead (not yet divided by sum_abs), to be
    finished by weighted_sums(part, ..., a_div=sum_abs, a_rows=rows)."""
    _need_gpu(P, T_db)
    assert P.dtype == _f32 and P.is_contiguous() and T_db.is_contiguous() and T_db.dtype == _f32
    batch, nframes, nfreq = P.shape
    rows = _rows(rows, batch, T_db.shape[0])
    if tuple(T_db.shape[1:]) != (nframes, nfreq) or (rows is None and T_db.shape[0] != batch) \
            or sum_abs.numel() != T_db.shape[0]:
        raise RuntimeError("edr_loss: target shape does not match the achieved EDR")
    lib = _lib.load()
    wf = None if wf is None else _f(wf)
    if defer:
        cols = lib.gfdn_edr_work_bytes(batch, nfreq) // (4 * batch)
        if out is not None and (out.dtype != _f32 or not out.is_contiguous() or tuple(out.shape) != (batch, cols)):
            raise RuntimeError("edr_loss(defer=True): out must be a contiguous float32 (batch, partial columns) tensor")
        part = torch.empty((batch, cols), dtype=_f32, device=P.device) if out is None else out
        _lib.check(lib.gfdn_edr_loss(_p(P), _p(T_db), _p(sum_abs), _p(rows), _p(wf), batch, nframes, nfreq,
                                     float(gscale), int(want_grad), None, _p(part), _stream()),
                   "gfdn_edr_loss")
        return part
    loss_item = torch.empty(batch, dtype=_f32, device=P.device)
    work = _work(lib.gfdn_edr_work_bytes(batch, nfreq), P.device)
    _lib.check(lib.gfdn_edr_loss(_p(P), _p(T_db), _p(sum_abs), _p(rows), _p(wf), batch, nframes, nfreq,
                                 float(gscale), int(want_grad), _p(loss_item), _p(work), _stream()),
               "gfdn_edr_loss")
    return loss_item


def edc_target(x, start: int, length: int) -> torch.Tensor:
    _need_gpu(x)
    x = _f(x)
    batch, ld = x.shape
    T_db = torch.empty((batch, length), dtype=_f32, device=x.device)
    lib = _lib.load()
    work = _work(lib.gfdn_edc_work_bytes(batch), x.device)
    _lib.check(lib.gfdn_edc_target(_p(x), ld, batch, start, length, _p(T_db), _p(work), _stream()),
               "gfdn_edc_target")
    return T_db


def edc_loss(x, start: int, length: int, T_db, maskw=None, inv_count: float = 1.0,
             gscale: float = 1.0, want_grad: bool = True, rows=None, item_len=None, items_per_band=None):
    """-> loss_item (batch,), gx (batch, ld) or None.  ``rows``: as in edr_loss.  ``item_len`` / ``items_per_band``: as in
    edc_loss_pairs (gfdn_edc_loss_banded)."""
    _need_gpu(x, T_db)
    x = _f(x)
    batch, ld = x.shape
    rows = _rows(rows, batch, T_db.shape[0])
    if T_db.dtype != _f32 or not T_db.is_contiguous() or (item_len is None and T_db.shape[-1] != length) \
            or (rows is None and T_db.shape[0] != batch):
        raise RuntimeError("edc_loss: target shape does not match the window")
    maskw = None if maskw is None else _f(maskw)
    loss_item = torch.empty(batch, dtype=_f32, device=x.device)
    gx = torch.empty_like(x) if want_grad else None
    lib = _lib.load()
    work = _work(lib.gfdn_edc_work_bytes(batch), x.device)
    if item_len is not None:
        ipb, ld_mask = _edc_bands(item_len, batch, items_per_band, maskw, length, T_db, "edc_loss")
        _lib.check(lib.gfdn_edc_loss_banded(_p(x), ld, batch, start, length, _p(item_len), _p(T_db), T_db.shape[-1],
                                            _p(rows), _p(maskw), ld_mask, ipb, float(inv_count), float(gscale),
                                            _p(loss_item), _p(gx), _p(work), _stream()), "gfdn_edc_loss_banded")
        return loss_item, gx
    _lib.check(lib.gfdn_edc_loss(_p(x), ld, batch, start, length, _p(T_db), _p(rows), _p(maskw),
                                 float(inv_count), float(gscale), _p(loss_item), _p(gx), _p(work),
                                 _stream()), "gfdn_edc_loss")
    return loss_item, gx


def draw_mask(seed: int, state, length: int, scale: float, out=None, band_len=None):
    """Fair-coin EDC time mask drawn on the device (counter-based; ``state`` is a 1-element int64
    step counter that the kernel advances) -> maskw (length,) = kept * scale / count.
    ``band_len`` (bands,) int32 device tensor: one row per band from the same bits, band q keeping the first
    band_len[q] <= length of them -> maskw (bands, length) (gfdn_draw_mask_banded)."""
    _need_gpu(state)
    if state.dtype != torch.int64 or state.numel() != 1:
        raise RuntimeError("draw_mask: state must be a 1-element int64 device tensor")
    lib = _lib.load()
    if band_len is not None:
        if band_len.dtype != torch.int32 or not band_len.is_contiguous() or not band_len.is_cuda:
            raise RuntimeError("draw_mask: band_len must be a contiguous int32 device tensor")
        nb = band_len.numel()
        if out is None:
            out = torch.empty((nb, length), dtype=_f32, device=state.device)
        if out.dtype != _f32 or tuple(out.shape) != (nb, length) or not out.is_contiguous():
            raise RuntimeError("draw_mask: out must be a contiguous float32 (bands, length) tensor")
        _lib.check(lib.gfdn_draw_mask_banded(int(seed) & 0xFFFFFFFFFFFFFFFF, _p(state), _p(band_len), nb, int(length),
                                             int(length), float(scale), _p(out), _stream()), "gfdn_draw_mask_banded")
        return out
    if out is None:
        out = torch.empty(length, dtype=_f32, device=state.device)
    _need_gpu(out)
    if out.dtype != _f32 or out.numel() != length or not out.is_contiguous():
        raise RuntimeError("draw_mask: out must be a contiguous float32 tensor of the mask length")
    _lib.check(lib.gfdn_draw_mask(int(seed) & 0xFFFFFFFFFFFFFFFF, _p(state), int(length), float(scale),
                                  _p(out), _stream()), "gfdn_draw_mask")
    return out


def mlp_gains_fwd(pos, freq_pi, w, H: int, n_hidden: int, G: int, lo: float, hi: float, rows=None,
                  nbands: int = 1, out=None):
    """pos (B,3) f64, freq_pi (F,) f32, w packed params -> gains (B,G), xhat (B,nl,H), rstd (B,nl).
    ``rows``: item b encodes pos[rows[b]] (pos = the positions of all receivers).
    ``nbands`` > 1: w (nbands, P), items band-major (item i uses parameter set i // (B / nbands)).
    ``out``: (gains, xhat, rstd) buffers to fill."""
    _need_gpu(pos, w)
    pos = pos.detach().to(torch.float64).contiguous()
    w = _f(w)
    B, F = (pos.shape[0] if rows is None else rows.numel()), freq_pi.numel()
    rows = _rows(rows, B, pos.shape[0])
    lib = _lib.load()
    if isinstance(H, (list, tuple)):
        # bands with their own layer sizes (csrc/mlp.hip k_mlp_bands_*): xhat / rstd are flat, band after band
        Hs, nh, sizes = _mlp_bands(H, n_hidden, nbands, B, F, G)
        if w.numel() != sizes[0]:
            raise RuntimeError("mlp_gains: packed parameter count does not match the bands' layer sizes")
        if out is not None:
            gains, xhat, rstd = out
            if tuple(gains.shape) != (B, G) or xhat.numel() != sizes[1] or rstd.numel() != sizes[2] \
                    or any(t.dtype != _f32 or not t.is_contiguous() for t in out):
                raise RuntimeError("mlp_gains_fwd: out = (gains (B, G), xhat, rstd) float32 contiguous of the bands' sizes")
        else:
            gains = torch.empty((B, G), dtype=_f32, device=pos.device)
            xhat = torch.empty(sizes[1], dtype=_f32, device=pos.device)
            rstd = torch.empty(sizes[2], dtype=_f32, device=pos.device)
        _lib.check(lib.gfdn_mlp_gains_bands_fwd(_p(pos), _p(rows), _p(freq_pi), _p(w), nbands, B // nbands, F, Hs, nh, G,
                                                float(lo), float(hi), _p(gains), _p(xhat), _p(rstd), _stream()),
                   "gfdn_mlp_gains_bands_fwd")
        return gains, xhat, rstd
    if w.numel() != nbands * lib.gfdn_mlp_param_count(F, H, n_hidden, G) or B % nbands:
        raise RuntimeError("mlp_gains: packed parameter count does not match the layer sizes")
    nl = 1 + n_hidden
    if out is not None:
        gains, xhat, rstd = out
        if tuple(gains.shape) != (B, G) or tuple(xhat.shape) != (B, nl, H) or tuple(rstd.shape) != (B, nl) \
                or any(t.dtype != _f32 or not t.is_contiguous() for t in out):
            raise RuntimeError("mlp_gains_fwd: out = (gains (B, G), xhat (B, nl, H), rstd (B, nl)) float32 contiguous")
    else:
        gains = torch.empty((B, G), dtype=_f32, device=pos.device)
        xhat = torch.empty((B, nl, H), dtype=_f32, device=pos.device)
        rstd = torch.empty((B, nl), dtype=_f32, device=pos.device)
    if nbands > 1:
        _lib.check(lib.gfdn_mlp_gains_banded_fwd(_p(pos), _p(rows), _p(freq_pi), _p(w), nbands, B // nbands, F, H,
                                                 n_hidden, G, float(lo), float(hi), _p(gains), _p(xhat),
                                                 _p(rstd), _stream()), "gfdn_mlp_gains_banded_fwd")
        return gains, xhat, rstd
    _lib.check(lib.gfdn_mlp_gains_fwd(_p(pos), _p(rows), _p(freq_pi), _p(w), B, F, H, n_hidden, G, float(lo),
                                      float(hi), _p(gains), _p(xhat), _p(rstd), _stream()),
               "gfdn_mlp_gains_fwd")
    return gains, xhat, rstd


_MLP_BANDS = {}


def _mlp_bands(H, n_hidden, nbands: int, B: int, F: int, G: int):
    """(H array, n_hidden array, sizes (w, xhat, rstd, backward work floats)) of a bank whose bands have their own layer
    sizes -- ctypes int arrays, cached per shape (the launches read them on the host)"""
    H, n_hidden = tuple(int(v) for v in H), tuple(int(v) for v in n_hidden)
    if len(H) != nbands or len(n_hidden) != nbands or B % nbands:
        raise RuntimeError("mlp_gains: one (H, n_hidden) per band, the same number of items per band")
    key = (H, n_hidden, B // nbands, F, G)
    if key not in _MLP_BANDS:
        Hs, nh = (ctypes.c_int * nbands)(*H), (ctypes.c_int * nbands)(*n_hidden)
        sizes = (ctypes.c_size_t * 4)()
        _lib.check(_lib.load().gfdn_mlp_bands_sizes(nbands, B // nbands, F, Hs, nh, G, sizes), "gfdn_mlp_bands_sizes")
        _MLP_BANDS[key] = (Hs, nh, tuple(int(v) for v in sizes))
    return _MLP_BANDS[key]


def mlp_bands_param_counts(H, n_hidden, F: int, G: int):
    """packed parameter count of every band's network"""
    lib = _lib.load()
    return [int(lib.gfdn_mlp_param_count(int(F), int(h), int(nh), int(G))) for h, nh in zip(H, n_hidden)]


def mlp_bwd_takes_parts(F: int, H, n_hidden, G: int, Bper: int) -> bool:
    """Whether mlp_gains_bwd(ggains_parts=...) applies to this network and batch (the wave-per-receiver kernel; always for
    the per-band sizes of k_mlp_bands_bwd, which sums the rows in either of its forms)."""
    if isinstance(H, (list, tuple)):
        return True
    return bool(_lib.load().gfdn_mlp_bwd_takes_parts(int(F), int(H), int(n_hidden), int(G), int(Bper)))


def mlp_gains_bwd(pos, freq_pi, w, H, n_hidden, G, lo, hi, gains, xhat, rstd, ggains, rows=None,
                  nbands: int = 1, out=None, ggains_parts=None, colscale=None):
    """-> gw shaped like w (``out``: a contiguous float32 buffer of that size to write it into).  ``ggains_parts`` (in
    place of ``ggains``): the (B G, chunks) partial rows of ``tf_gain_grad(partial=True)``, summed inside the launch."""
    _need_gpu(pos, w, ggains if ggains_parts is None else ggains_parts)
    pos = pos.detach().to(torch.float64).contiguous()
    w = _f(w)
    B, F = (pos.shape[0] if rows is None else rows.numel()), freq_pi.numel()
    rows = _rows(rows, B, pos.shape[0])
    lib = _lib.load()
    if out is not None and (out.dtype != _f32 or not out.is_contiguous() or out.numel() != w.numel()):
        raise RuntimeError("mlp_gains_bwd: out must be a contiguous float32 buffer of the parameter count")
    gw = torch.empty_like(w) if out is None else out
    if isinstance(H, (list, tuple)):
        Hs, nh, sizes = _mlp_bands(H, n_hidden, nbands, B, F, G)
        if w.numel() != sizes[0] or xhat.numel() != sizes[1] or rstd.numel() != sizes[2]:
            raise RuntimeError("mlp_gains_bwd: buffers do not match the bands' layer sizes")
        work = _work(4 * sizes[3], pos.device)
        if ggains_parts is not None:
            gsrc = _f(ggains_parts)
            if gsrc.dim() != 2 or gsrc.shape[0] != B * G:
                raise RuntimeError("mlp_gains_bwd: ggains_parts must be (B G, chunks)")
            gparts = gsrc.shape[1]
        else:
            gsrc, gparts = _f(ggains), 0
        _lib.check(lib.gfdn_mlp_gains_bands_bwd(_p(pos), _p(rows), _p(freq_pi), _p(w), nbands, B // nbands, F, Hs, nh, G,
                                                float(lo), float(hi), _p(gains), _p(xhat), _p(rstd), _p(gsrc), gparts,
                                                _p(colscale), _p(gw), _p(work), _stream()), "gfdn_mlp_gains_bands_bwd")
        return gw
    work = _work(lib.gfdn_mlp_bwd_work_bytes(B, F, H, n_hidden, G), pos.device)
    if ggains_parts is not None:
        gp = _f(ggains_parts)
        if gp.dim() != 2 or gp.shape[0] != B * G or B % nbands:
            raise RuntimeError("mlp_gains_bwd: ggains_parts must be (B G, chunks)")
        if colscale is not None:          # (the rows hold dL/d(gains colscale): tf_energy(gains=))
            _lib.check(lib.gfdn_mlp_gains_banded_bwd_parts_scaled(_p(pos), _p(rows), _p(freq_pi), _p(w), nbands, B // nbands, F,
                                                                  H, n_hidden, G, float(lo), float(hi), _p(gains), _p(xhat),
                                                                  _p(rstd), _p(gp), gp.shape[1], _p(colscale), _p(gw),
                                                                  _p(work), _stream()),
                       "gfdn_mlp_gains_banded_bwd_parts_scaled")
            return gw
        _lib.check(lib.gfdn_mlp_gains_banded_bwd_parts(_p(pos), _p(rows), _p(freq_pi), _p(w), nbands, B // nbands, F, H,
                                                       n_hidden, G, float(lo), float(hi), _p(gains), _p(xhat), _p(rstd),
                                                       _p(gp), gp.shape[1], _p(gw), _p(work), _stream()),
                   "gfdn_mlp_gains_banded_bwd_parts")
        return gw
    if colscale is not None:
        raise RuntimeError("mlp_gains_bwd: colscale goes with ggains_parts (the wave-per-receiver form)")
    ggains = _f(ggains)
    if nbands > 1:
        _lib.check(lib.gfdn_mlp_gains_banded_bwd(_p(pos), _p(rows), _p(freq_pi), _p(w), nbands, B // nbands, F, H,
                                                 n_hidden, G, float(lo), float(hi), _p(gains), _p(xhat),
                                                 _p(rstd), _p(ggains), _p(gw), _p(work), _stream()),
                   "gfdn_mlp_gains_banded_bwd")
        return gw
    _lib.check(lib.gfdn_mlp_gains_bwd(_p(pos), _p(rows), _p(freq_pi), _p(w), B, F, H, n_hidden, G, float(lo),
                                      float(hi), _p(gains), _p(xhat), _p(rstd), _p(ggains), _p(gw),
                                      _p(work), _stream()), "gfdn_mlp_gains_bwd")
    return gw


def adam_step(p, g, m, v, seg, lr_seg, step_count, beta1, beta2, eps, block_counter=None, mirror=None):
    """In-place fused Adam update of the flat buffers (all float32 on the GPU; seg uint8).  ``block_counter``: a
    zero-initialised int32 tensor of one element owned by the optimiser -- update and counter advance are then one
    launch for any size.  ``mirror``: a second counter that receives the advanced count too."""
    _need_gpu(p, g)
    if mirror is not None:
        _lib.check(_lib.load().gfdn_adam_step_mirrored(_p(p), _p(g), _p(m), _p(v), _p(seg), _p(lr_seg),
                                                       _p(step_count), _p(mirror), p.numel(), float(beta1),
                                                       float(beta2), float(eps), _p(block_counter), _stream()),
                   "gfdn_adam_step_mirrored")
        return
    if block_counter is not None:
        _lib.check(_lib.load().gfdn_adam_step_counted(_p(p), _p(g), _p(m), _p(v), _p(seg), _p(lr_seg),
                                                      _p(step_count), p.numel(), float(beta1), float(beta2),
                                                      float(eps), _p(block_counter), _stream()),
                   "gfdn_adam_step_counted")
        return
    _lib.check(_lib.load().gfdn_adam_step(_p(p), _p(g), _p(m), _p(v), _p(seg), _p(lr_seg),
                                          _p(step_count), p.numel(), float(beta1), float(beta2),
                                          float(eps), _stream()), "gfdn_adam_step")


# ------------------------------------------------------------------------------------------------
class KernelTimer:
    """HIP-event timing of ONE watched kernel on the launch stream (bench.py's roofline leg).

    ``watch`` names a kernel stage; wrappers that launch exactly that kernel bracket the launch
    with a pair of events on ``torch.cuda.current_stream()`` (the stream the C-ABI call launches
    on) while the timer is active.  Inactive (the default) it costs one attribute test."""

    def __init__(self):
        self.watch = None
        self.active = False
        self._events = []

    def start(self):
        self._events = []
        self.active = self.watch is not None

    def bracket(self, name: str, units: int):
        if self.active and (name == self.watch or (isinstance(self.watch, (tuple, set, frozenset)) and name in self.watch)):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            self._events.append((s, e, units, name))
            s.record()
            return e
        return None

    def stop_multi(self):
        """Several watched kernels (``watch`` = a tuple of names): {name: stats as stop()} plus, under 'window', the span from
        the first start to the last end over the k-th launches of all watched kernels (they run beside each other on
        different streams: the k-th launch of each belongs to the k-th step), averaged over the steps."""
        self.active = False
        if not self._events:
            return {}
        torch.cuda.synchronize()
        by = {}
        for s, e, u, name in self._events:
            by.setdefault(name, []).append((s, e, u))
        out = {}
        for name, evs in by.items():
            ms = [s.elapsed_time(e) for s, e, _ in evs]
            out[name] = {'kernel': name, 'launches': len(ms), 'avg_ms': sum(ms) / len(ms), 'min_ms': min(ms),
                         'units_per_launch': sum(u for _, _, u in evs) / len(evs)}
        names = list(by)
        n = min(len(v) for v in by.values())
        if len(names) > 1 and n > 0:
            spans = []
            for k in range(n):
                starts = [by[nm][k][0] for nm in names]
                ends = [by[nm][k][1] for nm in names]
                first = starts[0]
                for s_ in starts[1:]:
                    if s_.elapsed_time(first) > 0:          # (s_ was recorded before ``first``)
                        first = s_
                spans.append(max(first.elapsed_time(e_) for e_ in ends))
            out['window'] = {'kernels': names, 'avg_ms': sum(spans) / len(spans), 'min_ms': min(spans), 'steps': n}
        return out

    def stop(self):
        self.active = False
        if not self._events:
            return {}
        torch.cuda.synchronize()
        ms = [ev[0].elapsed_time(ev[1]) for ev in self._events]
        units = [ev[2] for ev in self._events]
        # what an event pair with NOTHING between its records measures on this stream: the part of
        # every bracket that is not the kernel (reported beside the raw figure, never hidden)
        empty = []
        for _ in range(32):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            e.record()
            empty.append((s, e))
        torch.cuda.synchronize()
        overhead = sorted(s.elapsed_time(e) for s, e in empty)[len(empty) // 2]
        return {'kernel': self.watch, 'launches': len(ms), 'avg_ms': sum(ms) / len(ms),
                'min_ms': min(ms), 'units_per_launch': sum(units) / len(units),
                'event_pair_overhead_ms': overhead}


kernel_timer = KernelTimer()


def pick_rows(table, state, idx):
    """idx <- table[state[0] mod state[1]]; state[0] += 1 (device-side; one launch).  table (len, B) int64."""
    _need_gpu(table, state, idx)
    if table.dtype != torch.long or state.dtype != torch.long or idx.dtype != torch.long or table.dim() != 2 \
            or table.shape[1] != idx.numel() or state.numel() != 2 or not table.is_contiguous():
        raise RuntimeError("pick_rows: table (len, B) int64 contiguous, state (2) int64, idx (B) int64")
    _lib.check(_lib.load().gfdn_pick_rows(_p(table), _p(state), _p(idx), idx.numel(), _stream()), "gfdn_pick_rows")
    return idx
