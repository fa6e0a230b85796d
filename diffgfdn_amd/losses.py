"""Energy-decay losses on the MI355X hot path.

Interface mirror of the reference's src/diff_gfdn/losses.py: ``edr_loss`` (:377-495),
``edc_loss`` (:149-281, broadband branch), ``directional_edc_loss`` (:284-371) with the same
constructor arguments and ``forward(target_response, achieved_response) -> 0-dim tensor``.

Pipeline per call (all HIP kernels, csrc/fft.hip + csrc/losses.hip):
    H (B,K) --irfft(n = K, odd: Bluestein)--> rir (B,K) --|STFT|^2--> EDR dB --> L1 ratio
                                                   \\--x^2 suffix-scan--> EDC dB --> mean |diff|
The loss is a scalar, so the backward of every stage is produced in the same call when the
achieved response requires grad and is handed to autograd as a saved tensor.  The target side
(EDR / EDC of the measured RIRs) does not depend on the model: it is computed once per distinct
target tensor and cached (:class:`DecayTargets`).

Off in every reference configuration: ERB grouping (``edr_loss(use_erb_grouping=True)``: supported at the
module level on the same kernels around two small matrix products, band matrix = input or the restated Slaney mel bank;
the trainers' fused steps do not take it), sub-band EDC (``band_centre_hz``) and ``reg_loss`` (NotImplementedError: both
need time-domain IIR filtering through learned / third-party second-order sections).
"""
import contextlib
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import hip_ops as ops
from .functional import IrfftPow2


def ms_to_samps(ms: float, fs: float) -> int:
    """int() truncation, reference utils.py:62-80."""
    return int(ms * 1e-3 * fs)


def edr_frequency_weights(sample_rate: float, win_size: int) -> torch.Tensor:
    """reference losses.py:419-428 with :49-57: 2 + (1 - 2) / (1 + exp(10^-2.5 (f - 1000)))
    (the reference passes `bottom, top` into the `top, bottom` slots; reproduced)."""
    freqs = torch.tensor(np.fft.rfftfreq(win_size, d=1.0 / sample_rate))
    return 2.0 + (1.0 - 2.0) / (1 + torch.exp(10 ** (-2.5) * (freqs - 1e3)))


def mel_frequencies(n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """Centre frequencies of the Slaney mel scale (librosa.mel_frequencies, htk=False -- the scale behind the
    reference's ``calc_erb_filters``, losses.py:18-46; librosa itself is not part of this build: restated from its
    published definition, parity unpinned): linear below 1 kHz (200 / 3 Hz per mel), logarithmic above (step ln 6.4 / 27)."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    return mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels))


def mel_filterbank(sample_rate: float, nfft: int, num_bands: int, fmin: float = 63.0, fmax: float = 16e3) -> np.ndarray:
    """(num_bands, nfft / 2 + 1) triangular filters with Slaney area normalisation (librosa.filters.mel defaults, the
    matrix the reference's ERB grouping multiplies |STFT| with, losses.py:34-38 and :545-551)."""
    fft_f = np.linspace(0.0, sample_rate / 2.0, 1 + nfft // 2)
    mel_f = mel_frequencies(num_bands + 2, fmin, fmax)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    return w * (2.0 / (mel_f[2:num_bands + 2] - mel_f[:num_bands]))[:, None]


def _as_batch(X: torch.Tensor) -> torch.Tensor:
    return X.reshape(-1, X.shape[-1])


def time_response_odd(X: torch.Tensor) -> torch.Tensor:
    """rir = torch.fft.irfft(X, n = X.shape[-1])  (reference losses.py:207-213, :442-445)."""
    n = X.shape[-1]
    if n % 2 == 0:
        raise NotImplementedError("irfft(X, n=K) with even K is not produced by the reference's grids")
    return ops.irfft_odd_fwd(_as_batch(X), n)


class DecayTargets:
    """EDR / EDC of the TARGET responses, computed once and cached.

    A target tensor is identified by (data_ptr, shape, version).  ``edr``: (T_db (B,frames,F),
    sum_abs (B,)); ``edc``: T_db (B, len)."""

    def __init__(self, max_entries: int = 64):
        self._rir: Dict = {}
        self._edr: Dict = {}
        self._edc: Dict = {}
        self.max_entries = max_entries

    @staticmethod
    def _key(t: torch.Tensor, *extra):
        return (t.data_ptr(), tuple(t.shape), t._version, str(t.device)) + extra

    def _trim(self, d: Dict):
        if len(d) >= self.max_entries:
            d.clear()

    # every entry keeps a reference to its key tensor: the storage cannot be freed and re-used
    # by another tensor while the entry is alive, so (data_ptr, version) identifies the contents
    def rir(self, target: torch.Tensor) -> torch.Tensor:
        k = self._key(target)
        r = self._rir.get(k)
        if r is None:
            self._trim(self._rir)
            r = (time_response_odd(target), target)
            self._rir[k] = r
        return r[0]

    def edr(self, target: torch.Tensor, win: int):
        k = self._key(target, win)
        e = self._edr.get(k)
        if e is None:
            self._trim(self._edr)
            P = ops.stft_power(self.rir(target), win)
            e = (ops.edr_target(P), target)
            self._edr[k] = e
        return e[0]

    def edc(self, target: torch.Tensor, start: int, length: int):
        k = self._key(target, start, length)
        e = self._edc.get(k)
        if e is None:
            self._trim(self._edc)
            e = (ops.edc_target(self.rir(target), start, length), target)
            self._edc[k] = e
        return e[0]

    def drop_rirs(self):
        self._rir.clear()


_default_targets = DecayTargets()


# ``unit_grad=True`` promises that the loss enters the final scalar with weight 1 (the saved dloss/dH is handed on as it
# is, one pass over it saved).  The promise is CHECKED on EVERY backward call, without a host read: a one-element compare on
# the device ORs "the upstream gradient was not 1" into a per-device flag (captured into HIP graphs like any other launch, so
# replayed steps are checked too), and the flag is read where the host synchronises anyway -- the trainers' epoch ends,
# ``raise_on_unit_grad_violation()`` -- so a caller that rescales the total (gradient accumulation, loss scaling, a 1 / world
# factor) gets an error instead of an unscaled EDC gradient next to correctly scaled colorless gradients.  (Round 5 read the
# device inside backward -- a host sync that drains the side streams -- and therefore checked the first eight calls of a
# process only: a second trainer, or one that enabled a loss scale later, was never checked.)
_UNIT_GRAD_FLAGS = {}


def _assert_unit_upstream(g: torch.Tensor):
    if not g.is_cuda:
        return
    key = g.device.index
    flag = _UNIT_GRAD_FLAGS.get(key)
    if flag is None:
        if torch.cuda.is_current_stream_capturing():
            return                    # (first call inside a capture: no allocation there; the warm-up steps ran before it)
        flag = _UNIT_GRAD_FLAGS[key] = torch.zeros((), dtype=torch.bool, device=g.device)
    flag.logical_or_((g != 1).any())


def raise_on_unit_grad_violation():
    """Reads the per-device flags (a host sync: call where the host waits for the device anyway) and raises if a decay loss
    evaluated with ``unit_grad=True`` was back-propagated with an upstream gradient other than 1 since the last call."""
    for key, flag in _UNIT_GRAD_FLAGS.items():
        if bool(flag):
            flag.zero_()
            raise RuntimeError("decay losses were evaluated with unit_grad=True but their total is back-propagated with an "
                               "upstream gradient other than 1: evaluate them with unit_grad=False to rescale the total")


class _ScalarLossWithSavedGrad(torch.autograd.Function):
    """loss value + precomputed dloss/dH (saved) -> autograd node."""

    @staticmethod
    def forward(ctx, H, loss_value, gH, unit_grad=False):
        ctx.save_for_backward(gH)
        ctx.h_shape = H.shape
        ctx.unit_grad = unit_grad
        return loss_value.clone()

    @staticmethod
    def backward(ctx, g):
        (gH,) = ctx.saved_tensors
        if ctx.unit_grad:            # caller guarantees d(total)/d(loss) == 1: skip a pass over gH
            _assert_unit_upstream(g)
            return gH.reshape(ctx.h_shape), None, None, None
        return (gH * g).reshape(ctx.h_shape), None, None, None


class _ScalarLossWithSavedGrads(torch.autograd.Function):
    """loss value + precomputed gradients of several inputs (saved) -> autograd node."""

    @staticmethod
    def forward(ctx, loss_value, unit_grad, *inputs_and_grads):
        k = len(inputs_and_grads) // 2
        ctx.save_for_backward(*inputs_and_grads[k:])
        ctx.meta = [(t.shape, t.dtype) for t in inputs_and_grads[:k]]
        ctx.unit_grad = unit_grad
        return loss_value.clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.unit_grad:
            _assert_unit_upstream(g)
            out = [t.to(dt).reshape(sh) for t, (sh, dt) in zip(ctx.saved_tensors, ctx.meta)]
        else:
            out = [(t * g).to(dt).reshape(sh) for t, (sh, dt) in zip(ctx.saved_tensors, ctx.meta)]
        return (None, None, *out, *([None] * len(out)))


class _DecayTotal(torch.autograd.Function):
    """[w_edr sum(edr_items) + w_edc sum(edc_items), w_edr sum(.), w_edc sum(.)] as three 0-dim outputs
    of ONE bookkeeping launch, the first carrying the precomputed dloss/dH.  The outputs are created
    inside the node: no clone, no select-backward (zero fill + copy) on the way to the loss."""

    @staticmethod
    def forward(ctx, H, gH, unit_grad, li_edr, w_edr, li_edc, w_edc, edr_div=None, edr_rows=None,
                nbands: int = 1):
        sums = ops.weighted_sums(li_edr, w_edr, li_edc, w_edc, edr_div, edr_rows, nbands)
        ctx.set_materialize_grads(False)      # no zero-filled gradients for the two report outputs
        ctx.save_for_backward(gH)
        ctx.h_shape = H.shape
        ctx.unit_grad = unit_grad
        if nbands > 1:
            if not unit_grad:
                raise ValueError("band-stacked decay losses are back-propagated with unit gradients")
            total, edr, edc = sums[:, 0], sums[:, 1], sums[:, 2]      # (nbands,) per-band sums
        else:
            total, edr, edc = sums[0], sums[1], sums[2]
        ctx.mark_non_differentiable(edr, edc)
        return total, edr, edc

    @staticmethod
    def backward(ctx, g, _g1, _g2):
        (gH,) = ctx.saved_tensors
        if g is None:
            return (None,) * 10
        if ctx.unit_grad:
            _assert_unit_upstream(g)
            return (gH.reshape(ctx.h_shape),) + (None,) * 9
        return ((gH * g).reshape(ctx.h_shape),) + (None,) * 9


def shard_loss_scales(world_size: int, global_batch: int, mask_count: float) -> Dict[str, float]:
    """What a rank multiplies its LOCAL loss sums with so that ONE all-reduce (sum) of gradients and loss slots over the
    ranks equals the single-process step on the whole batch (reference losses.py:478-492, :235-238,
    trainer.py:298-308):
      'edr'       -- the EDR term is a SUM over items of per-item ratios: local sums add up as they are;
      'edc'       -- the EDC term is a MEAN over (global batch x kept time indices): every local |dB| term carries
                     1 / (global_batch * mask_count), the same time mask on every rank;
      'colorless' -- spectral and sparsity terms do not depend on the receiver: every rank evaluates them in full and
                     contributes 1 / world_size of them.
    The trainers (VarReceiverPosTrainer._step_losses, bankstep.FusedBankStep.run, decay_losses) take their factors from
    here; tests/test_distributed_cpu.py drives the same function with the CPU oracle as the compute."""
    if world_size < 1 or global_batch < 1 or mask_count <= 0:
        raise ValueError("shard_loss_scales: world_size, global_batch >= 1 and mask_count > 0")
    return {'edr': 1.0, 'edc': 1.0 / (float(global_batch) * float(mask_count)), 'colorless': 1.0 / float(world_size)}


def decay_losses(H: torch.Tensor, target: Optional[torch.Tensor] = None, *, win: int = 4096,
                 edr_weight: float = 1.0, edc_weight: float = 1.0, use_edr: bool = True,
                 use_edc: bool = True, edc_start: int = 640, edc_len: Optional[int] = None,
                 edc_maskw: Optional[torch.Tensor] = None, edc_count: Optional[float] = None,
                 edc_maskw_prenormalised: bool = False,
                 freq_weights: Optional[torch.Tensor] = None,
                 reduced_pole_radius: Optional[float] = None,
                 global_batch: Optional[int] = None,
                 targets: Optional[DecayTargets] = None,
                 edr_target: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                 edc_target: Optional[torch.Tensor] = None,
                 side_stream: Optional["torch.cuda.Stream"] = None,
                 unit_grad: bool = False, n_time: Optional[int] = None,
                 target_rows: Optional[torch.Tensor] = None, nbands: int = 1, slot_order: bool = False,
                 pairs: bool = False, join_event=None, edc_item_len: Optional[torch.Tensor] = None,
                 edc_items_per_band: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Fused EDR + EDC evaluation sharing ONE irfft of H and ONE adjoint transform.

    Returns (total, w_edr * edr, w_edc * edc) with total = their sum carrying the gradient; the two
    parts are detached values (trainer.py:280-288 logs them separately).
    ``global_batch``: number of items the EDC mean runs over (the local batch unless the batch
    is sharded over ranks).  ``edr_target`` = (T_db, sum_abs) / ``edc_target`` = T_db
    may be passed precomputed (dataset-level store); otherwise they are derived from ``target``
    through the cache.  ``side_stream``: run the EDC kernel (one block per item, latency-bound)
    beside the STFT -> EDR chain instead of in front of it.  ``unit_grad``: the caller guarantees
    that the returned total enters the final loss with weight 1 (skips one rescale of dL/dH).
    The returned ``edr`` / ``edc`` are the WEIGHTED parts.  ``n_time``: length of the time response
    (= number of bins K of the full grid); H may then hold only the (K+1)/2 bins the transform
    irfft(X, n = K) actually reads.  ``target_rows``: int64 index; ``edr_target`` / ``edc_target``
    are then stores over ALL receivers and item b compares against row target_rows[b].
    ``nbands`` > 1: the items are band-major batches of ``nbands`` independent models (BandBank); the
    three results are then (nbands,) vectors of per-band sums (``global_batch`` = items per band).
    ``slot_order``: H was evaluated on the slot-ordered grid of ``ops.irfft_slot_order(n_time)`` (column 0 =
    bin 0, column 1 + s = slot s): the transform then needs no gather and dL/dH comes back in the same order.
    ``pairs`` (with slot_order, win 4096, precomputed targets): two items ride one complex transform and the time
    signals stay pair-interleaved (float2) through the STFT / EDC kernels: one 8-byte scatter / gather per slot
    serves two items and the transform passes move half the work blocks.
    ``edc_item_len`` (items,) int32 + ``edc_items_per_band``: per-item EDC window lengths <= ``edc_len`` (band banks whose
    bands differ in T60max; ``edc_target`` rows padded to ``edc_len``, ``edc_maskw`` (bands, edc_len) pre-normalised)."""
    if edc_item_len is not None and not (edc_maskw_prenormalised and edc_maskw is not None):
        raise ValueError("per-item EDC windows take pre-normalised per-band weight rows (every band has its own count)")
    if pairs:
        return _decay_losses_pairs(H, win=win, edr_weight=edr_weight, edc_weight=edc_weight, edc_start=edc_start,
                                   edc_len=edc_len, edc_maskw=edc_maskw, edc_count=edc_count,
                                   edc_maskw_prenormalised=edc_maskw_prenormalised, global_batch=global_batch,
                                   edr_target=edr_target, edc_target=edc_target, side_stream=side_stream,
                                   unit_grad=unit_grad, n_time=n_time, target_rows=target_rows, nbands=nbands,
                                   slot_order=slot_order, freq_weights=freq_weights,
                                   reduced_pole_radius=reduced_pole_radius, join_event=join_event,
                                   edc_item_len=edc_item_len, edc_items_per_band=edc_items_per_band)
    if target_rows is not None and (edr_target is None and use_edr or edc_target is None and use_edc):
        raise ValueError("target_rows needs precomputed target stores")
    targets = targets or _default_targets
    Hb = _as_batch(H)
    B, ldx = Hb.shape
    K = ldx if n_time is None else n_time
    if ldx < (K + 1) // 2:
        raise ValueError("H holds fewer bins than irfft(X, n) reads")
    want_grad = H.requires_grad and torch.is_grad_enabled()
    x = ops.irfft_odd_fwd(Hb, K, slots=slot_order)
    env = None
    if reduced_pole_radius is not None and reduced_pole_radius != 1.0:
        # losses.py:447-451: undo sampling on a larger circle (EDR only in the reference)
        env = torch.pow(torch.tensor(1.0 / reduced_pole_radius, dtype=torch.float64, device=x.device),
                        torch.arange(K, device=x.device, dtype=torch.float64)).to(torch.float32)
    gx = gx2 = None
    li_edc = li_edr = edr_div = edr_rows = None
    main = torch.cuda.current_stream() if x.is_cuda else None
    fork = side_stream is not None and use_edc and use_edr
    if use_edc:
        L = edc_len if edc_len is not None else K - edc_start
        T_db = edc_target if edc_target is not None else targets.edc(target, edc_start, L)
        count = float(L) if edc_count is None else float(edc_count)
        nb = B // nbands if global_batch is None else global_batch
        # pre-normalised weights already carry 1 / (items * kept indices): no host scalar varies
        # from step to step, which keeps the launch arguments static under graph replay
        inv = 1.0 if edc_maskw_prenormalised else shard_loss_scales(1, nb, count)['edc']
        if fork:
            side_stream.wait_stream(main)
            with torch.cuda.stream(side_stream):
                li_edc, gx = ops.edc_loss(x, edc_start, L, T_db, edc_maskw, inv, edc_weight, want_grad,
                                          rows=target_rows, item_len=edc_item_len, items_per_band=edc_items_per_band)
                x.record_stream(side_stream)
        else:
            li_edc, gx = ops.edc_loss(x, edc_start, L, T_db, edc_maskw, inv, edc_weight, want_grad,
                                      rows=target_rows, item_len=edc_item_len, items_per_band=edc_items_per_band)
    if use_edr:
        T_edr, sum_abs = edr_target if edr_target is not None else targets.edr(target, win)
        xe = x if env is None else x * env
        # the STFT adjoint scatters exactly two frame terms per sample with atomic adds into a
        # ZEROED buffer (a + b commutes -> order independent, bitwise reproducible), cleared by
        # the forward STFT launch; the adjoint irfft then reads EDC gradient + this buffer, in
        # that order, so the three-term sum has a fixed order too
        g_edr = torch.empty_like(x) if want_grad else None
        P = ops.stft_power(xe, win, zero_buf=g_edr)
        # (the per-item reduction of the EDR partials is deferred into the bookkeeping launch)
        li_edr = ops.edr_loss(P, T_edr, sum_abs, freq_weights, edr_weight, want_grad, rows=target_rows,
                              defer=True)
        edr_div, edr_rows = sum_abs, target_rows
        if want_grad:
            g_edr = ops.stft_power_bwd(xe, win, P, g_edr)
            if env is not None:
                g_edr = g_edr * env
        if fork:
            main.wait_stream(side_stream)
            for t in (gx, li_edc):
                if t is not None:
                    t.record_stream(main)
        if want_grad:
            gx, gx2 = (g_edr, None) if gx is None else (gx, g_edr)
    if want_grad:
        gH = ops.irfft_odd_bwd(gx, K, ldx, gx2, slots=slot_order)
        return _DecayTotal.apply(H, gH, unit_grad, li_edr, edr_weight, li_edc, edc_weight, edr_div, edr_rows,
                                 nbands)
    sums = ops.weighted_sums(li_edr, edr_weight, li_edc, edc_weight, edr_div, edr_rows, nbands)   # [total, w_edr edr, w_edc edc]
    if nbands > 1:
        return sums[:, 0], sums[:, 1], sums[:, 2]
    return sums[0], sums[1], sums[2]


def _decay_losses_pairs(H, *, win, edr_weight, edc_weight, edc_start, edc_len, edc_maskw, edc_count,
                        edc_maskw_prenormalised, global_batch, edr_target, edc_target, side_stream, unit_grad,
                        n_time, target_rows, nbands, slot_order, freq_weights, reduced_pole_radius, join_event=None,
                        edc_item_len=None, edc_items_per_band=None):
    """decay_losses on pair-interleaved time signals (see there); EDR and EDC both on, targets precomputed.
    ``join_event``: an event of another branch of the step that the loss kernels wait for behind the transform
    (scheduling only: it pulls that branch in front of the loss kernels instead of beside their adjoints)."""
    if not slot_order or win != 4096 or edr_target is None or edc_target is None or n_time is None:
        raise ValueError("pairs: slot-ordered H, win 4096 and precomputed EDR / EDC targets are required")
    if reduced_pole_radius is not None and reduced_pole_radius != 1.0:
        raise NotImplementedError("pairs: unit-circle sampling only")
    Hb = _as_batch(H)
    B = Hb.shape[0]
    K = n_time
    want_grad = H.requires_grad and torch.is_grad_enabled()
    x2 = ops.irfft_odd_fwd(Hb, K, slots=True, pairs=True)
    main = torch.cuda.current_stream()
    if join_event is not None:
        main.wait_event(join_event)
    L = edc_len if edc_len is not None else K - edc_start
    count = float(L) if edc_count is None else float(edc_count)
    nb = B // nbands if global_batch is None else global_batch
    inv = 1.0 if edc_maskw_prenormalised else 1.0 / (nb * count)
    fork = side_stream is not None
    if fork:
        side_stream.wait_stream(main)
    with torch.cuda.stream(side_stream) if fork else contextlib.nullcontext():
        li_edc, g_edc = ops.edc_loss_pairs(x2, B, edc_start, L, edc_target, edc_maskw, inv, edc_weight, want_grad,
                                           rows=target_rows, item_len=edc_item_len, items_per_band=edc_items_per_band)
        if fork:
            x2.record_stream(side_stream)
    T_edr, sum_abs = edr_target
    P = ops.stft_power_pairs(x2, B, win)
    li_edr = ops.edr_loss(P, T_edr, sum_abs, freq_weights, edr_weight, want_grad, rows=target_rows, defer=True)
    if fork:
        main.wait_stream(side_stream)
        for t in (g_edc, li_edc):
            if t is not None:
                t.record_stream(main)
    if want_grad:
        # the STFT adjoint stores EDC gradient + its own contribution (two non-overlapping frame sets, no atomics)
        # into the EDC buffer: one gradient signal for the transform's adjoint to gather
        g = ops.stft_power_pairs_bwd(x2, B, win, P, base=g_edc, out=g_edc)
        gH = ops.irfft_odd_pairs_bwd(g, K, B)
        return _DecayTotal.apply(H, gH, unit_grad, li_edr, edr_weight, li_edc, edc_weight, sum_abs, target_rows, nbands)
    sums = ops.weighted_sums(li_edr, edr_weight, li_edc, edc_weight, sum_abs, target_rows, nbands)
    if nbands > 1:
        return sums[:, 0], sums[:, 1], sums[:, 2]
    return sums[0], sums[1], sums[2]


class edr_loss(nn.Module):
    """Difference between the EDRs of two RIRs in dB (reference losses.py:377-495)."""

    def __init__(self, sample_rate: float, win_size: int = 2 ** 12, hop_size: int = 2 ** 11,
                 reduced_pole_radius: Optional[float] = None, use_erb_grouping: bool = False,
                 time_axis: int = -1, freq_axis: int = -2, use_weight_fn: bool = False,
                 erb_filters=None):
        """``use_erb_grouping``: the EDR is taken on |STFT| grouped into 64 bands (reference losses.py:545-551).  The
        reference builds the band matrix with librosa (absent here): pass it as ``erb_filters`` (bands, win / 2 + 1), or
        leave it to the restated Slaney mel filterbank (`mel_filterbank`, parity unpinned)."""
        super().__init__()
        if hop_size * 2 != win_size:
            raise NotImplementedError("the reference asserts hop == win // 2 (losses.py:524)")
        self.sample_rate = sample_rate
        self.win_size = win_size
        self.hop_size = hop_size
        self.use_erb_grouping = use_erb_grouping
        self.reduced_pole_radius = reduced_pole_radius
        self.time_axis = time_axis
        self.freq_axis = freq_axis
        self.use_weight_fn = use_weight_fn
        self.erb_filters = None
        self.freqs_hz = np.fft.rfftfreq(win_size, d=1.0 / sample_rate)
        if use_erb_grouping:
            if erb_filters is None:
                erb_filters = mel_filterbank(sample_rate, win_size, 2 ** 6)
                self.freqs_hz = mel_frequencies(2 ** 6, 63.0, 16e3)
            self.erb_filters = torch.as_tensor(np.asarray(erb_filters), dtype=torch.float32)
            if self.erb_filters.shape[-1] != win_size // 2 + 1:
                raise ValueError("erb_filters must be (bands, win_size / 2 + 1)")
        if use_weight_fn:
            if use_erb_grouping:
                if len(self.freqs_hz) != self.erb_filters.shape[0]:
                    raise ValueError("frequency weighting of a caller-supplied band matrix needs its band frequencies")
                f = torch.tensor(self.freqs_hz)
                self.frequency_weights = 2.0 + (1.0 - 2.0) / (1 + torch.exp(10 ** (-2.5) * (f - 1e3)))
            else:
                self.frequency_weights = edr_frequency_weights(sample_rate, win_size)
        self.targets = _default_targets

    def forward(self, target_response: torch.Tensor, achieved_response: torch.Tensor) -> torch.Tensor:
        assert target_response.shape == achieved_response.shape
        wf = self.frequency_weights.to(achieved_response.device) if self.use_weight_fn else None
        if self.use_erb_grouping:
            return self._forward_erb(target_response, achieved_response, wf)
        total, _, _ = decay_losses(achieved_response, target_response, win=self.win_size,
                                   use_edc=False, freq_weights=wf,
                                   reduced_pole_radius=self.reduced_pole_radius,
                                   targets=self.targets)
        return total


    def _forward_erb(self, target_response: torch.Tensor, achieved_response: torch.Tensor, wf) -> torch.Tensor:
        """EDR loss on |STFT| grouped into bands, S_band = erb_filters |S| (losses.py:545-551), EDR = tail sums of S_band^2
        (:556-575): the transform, STFT, EDR and loss kernels of the ungrouped path around two small matrix products; the
        gradient is assembled in the forward and handed to autograd as the saved dloss/dH."""
        Hb, Tb = _as_batch(achieved_response), _as_batch(target_response)
        K, win = Hb.shape[-1], self.win_size
        erb = self.erb_filters.to(Hb.device)
        want_grad = achieved_response.requires_grad and torch.is_grad_enabled()
        env = None
        if self.reduced_pole_radius is not None and self.reduced_pole_radius != 1.0:
            env = torch.pow(torch.tensor(1.0 / self.reduced_pole_radius, dtype=torch.float64, device=Hb.device),
                            torch.arange(K, device=Hb.device, dtype=torch.float64)).to(torch.float32)

        def grouped(X, e):
            x = ops.irfft_odd_fwd(X, K)
            if e is not None:
                x = x * e
            mag = ops.stft_power(x, win).sqrt_()                          # (B, frames, F) |STFT|
            E = mag @ erb.T                                               # (B, frames, bands)
            return x, mag, E

        with torch.no_grad():
            _, _, Et = grouped(Tb.to(torch.complex64), None)
            T_db, sum_abs = ops.edr_target((Et * Et).contiguous())
            x, mag, E = grouped(Hb.detach().to(torch.complex64), env)
            gPe = (E * E).contiguous()
            li = ops.edr_loss(gPe, T_db, sum_abs, wf, 1.0, want_grad)     # gPe becomes dloss/d(E^2)
            val = li.sum()
            if not want_grad:
                return val
            gmag = (2.0 * E * gPe) @ erb                                  # d/d|S|
            gP = torch.where(mag > 0, 0.5 * gmag / mag, torch.zeros_like(mag)).contiguous()
            gx = ops.stft_power_bwd(x, win, gP, torch.zeros_like(x))
            if env is not None:
                gx = gx * env
            gH = ops.irfft_odd_bwd(gx, K, K)
        return _ScalarLossWithSavedGrad.apply(achieved_response, val, gH.reshape(achieved_response.shape), False)


class edc_loss(nn.Module):
    """Broadband EDC loss in dB (reference losses.py:149-281)."""

    def __init__(self, max_ir_len_ms: float, sample_rate: float, band_centre_hz: Optional[List] = None,
                 mixing_time_ms: float = 20.0, use_mask: bool = False):
        super().__init__()
        if band_centre_hz is not None:
            raise NotImplementedError("sub-band EDC is never enabled by the trainer (trainer.py:81-83)")
        self.max_ir_len_samps = ms_to_samps(max_ir_len_ms, sample_rate)
        self.band_centre_hz = band_centre_hz
        self.mixing_time_samps = ms_to_samps(mixing_time_ms, sample_rate)
        self.use_mask = use_mask
        self.targets = _default_targets

    def window(self, K: int) -> Tuple[int, int]:
        L = min(self.max_ir_len_samps, K)
        return self.mixing_time_samps, L - self.mixing_time_samps

    def draw_mask(self, length: int, device) -> Tuple[Optional[torch.Tensor], float]:
        """Random time mask of losses.py:221-223, drawn from the global CPU generator exactly as
        the reference does (uniform_ then bernoulli) so seeded runs see the same indices."""
        if not self.use_mask:
            return None, float(length)
        probs = torch.empty(length).uniform_(0, 1)
        keep = torch.bernoulli(probs)
        return keep.to(device=device, dtype=torch.float32), float(keep.sum().item())

    def forward(self, target_response: torch.Tensor, achieved_response: torch.Tensor,
                mask_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        K = achieved_response.shape[-1]
        start, length = self.window(K)
        if mask_index is not None:
            maskw = torch.zeros(length, dtype=torch.float32, device=achieved_response.device)
            maskw[mask_index.reshape(-1).long().to(maskw.device)] = 1.0
            count = float(mask_index.numel())
        else:
            maskw, count = self.draw_mask(length, achieved_response.device)
        total, _, _ = decay_losses(achieved_response, target_response, use_edr=False,
                                   edc_start=start, edc_len=length, edc_maskw=maskw,
                                   edc_count=count, targets=self.targets)
        return total


class directional_edc_loss(nn.Module):
    """Mean |dB| between the EDCs of predicted directional RIRs and a common-slope model
    (reference losses.py:284-371).

    The reference builds ``envelopes`` (slopes x time) with slope2noise.utils.decay_kernel, an
    un-vendored dependency: pass them as ``envelopes``.  When omitted, the stated formula
    exp(-13.8155 t / T60) (EDC of an exponential decay, unit at t = 0) is used; parity for that
    formula is unpinned."""

    def __init__(self, common_decay_times, edc_len_ms: float, fs: float, mixing_time_ms: float = 20.0,
                 use_mask: bool = False, envelopes: Optional[torch.Tensor] = None):
        super().__init__()
        self.mixing_time_samps = ms_to_samps(mixing_time_ms, fs)
        self.use_mask = use_mask
        self.edc_len_samps = ms_to_samps(edc_len_ms, fs)
        if envelopes is None:
            cdt = np.asarray(common_decay_times, dtype=np.float64).reshape(-1)
            t = np.linspace(0, (self.edc_len_samps - 1) / fs, self.edc_len_samps)
            envelopes = torch.tensor(np.exp(-13.815510557964274 * t[None, :] / cdt[:, None]),
                                     dtype=torch.float32)
        self.register_buffer('envelopes', torch.as_tensor(envelopes, dtype=torch.float32),
                             persistent=False)
        # (seed, state, maskw) of a device-side mask generator (ops.draw_mask): when set, every masked evaluation draws its
        # time mask on the device -- no host read, so the step can be captured into a HIP graph (GraphedModuleStep)
        self.device_mask = None

    def _time_mask(self, L: int, device):
        """(maskw or None, what the kernel divides by beyond the weights): losses.py:355-360 draws
        argwhere(bernoulli(U(0, 1))) over the window on the host; with ``device_mask`` the same fair bits come from the
        counter-based device generator and the weights arrive divided by the kept count."""
        if not self.use_mask:
            return None, float(L)
        if self.device_mask is not None:
            seed, state, maskw = self.device_mask
            if maskw.numel() != L:
                raise ValueError("device_mask: the weight buffer does not match the EDC window")
            ops.draw_mask(seed, state, L, 1.0, out=maskw)
            return maskw, 1.0
        keep = torch.bernoulli(torch.empty(L).uniform_(0, 1))
        return keep.to(device=device, dtype=torch.float32), float(keep.sum().item())

    def forward(self, H_pred: torch.Tensor, amps_true: torch.Tensor, weight: float = 1.0,
                unit_grad: bool = False) -> torch.Tensor:
        """H_pred (B, J, K) complex, amps_true (B, J, S) -> ``weight`` x the loss (reference: weight 1).
        ``unit_grad``: the caller guarantees that the returned value enters the total with factor 1 -- the weight then
        rides the EDC kernel's gradient scale and the backward hands the saved dL/dH on as it is (no pass over the
        (B J, K) gradient to multiply it by the upstream scalar: 2 x 201 MB at 32 receivers x 12 directions)."""
        B, J, K = H_pred.shape
        n = 2 * (K - 1)
        # python slicing in the reference (:344-346) silently truncates at the end of the IR
        start = self.mixing_time_samps
        L = min(self.edc_len_samps, n - start)
        if L <= 0:
            raise ValueError("EDC window starts beyond the impulse-response length")
        want_grad = H_pred.requires_grad and torch.is_grad_enabled()
        Hb = H_pred.reshape(B * J, K)
        x = ops.irfft_pow2_fwd(Hb, n)
        # true EDC from the common-slope amplitudes (einsum 'bjk,kt->bjt'), then dB
        if self.envelopes.device != x.device:              # (once: a host -> device copy cannot be graph-captured)
            self.envelopes = self.envelopes.to(x.device)
        maskw, count = self._time_mask(L, x.device)
        # (the einsum 'bjk,kt->bjt' of the amplitudes with the envelopes, |.| + eps, dB and the clip at -200 happen
        # inside the EDC scan: the (B J, L) target and the six passes over it that built it never exist)
        amps = amps_true.to(device=x.device, dtype=torch.float32).reshape(B * J, -1).contiguous()
        li, gx = ops.edc_loss_model(x, start, L, amps, self.envelopes, maskw, 1.0 / (B * J * count), float(weight),
                                    want_grad)
        val = li.sum() if weight == 1.0 else li.sum() * float(weight)
        if not want_grad:
            return val
        gH = ops.irfft_pow2_bwd(gx, n).reshape(B, J, K)
        return _ScalarLossWithSavedGrad.apply(H_pred, val, gH, unit_grad)

    def forward_sh(self, H_sh: torch.Tensor, analysis_matrix: torch.Tensor, amps_true: torch.Tensor,
                   weight: float = 1.0, unit_grad: bool = False) -> torch.Tensor:
        """``forward(einsum('jl,blk->bjk', A, H_sh), amps_true)`` (reference trainer.py:853-865 followed by
        losses.py:333-371) with the two linear maps in the other order: the inverse transform runs on the C
        SH-domain responses of a receiver and the analysis matrix mixes the TIME signals into the J directions (A is
        real: irfft(A H) = A irfft(H)) -- C instead of J transforms per receiver each way (9 instead of 12 at order 2),
        and the (B, J, K) directional spectra and their gradient never exist.  H_sh (B, C, K) complex, A (J, C)."""
        B, C, K = H_sh.shape
        J = analysis_matrix.shape[0]
        n = 2 * (K - 1)
        start = self.mixing_time_samps
        L = min(self.edc_len_samps, n - start)
        if L <= 0:
            raise ValueError("EDC window starts beyond the impulse-response length")
        want_grad = H_sh.requires_grad and torch.is_grad_enabled()
        x_sh = ops.irfft_pow2_fwd(H_sh.reshape(B * C, K), n)                       # (B C, n)
        if self.envelopes.device != x_sh.device:
            self.envelopes = self.envelopes.to(x_sh.device)
        amps = amps_true.to(device=x_sh.device, dtype=torch.float32).reshape(B * J, -1).contiguous()
        if n == 131072 and ops.edc_mixed_supported(C, J, amps.shape[1]):
            # the J directional samples formed in registers inside the EDC kernels (csrc/edcmix.hip): neither the
            # directional signals nor their gradient exist, the adjoint transform reads the EDC window only
            maskw, count = self._time_mask(L, x_sh.device)
            li, gx_sh = ops.edc_loss_model_mixed(x_sh.view(B, C, n), analysis_matrix, start, L, amps, self.envelopes, maskw,
                                                 1.0 / (B * J * count), float(weight), want_grad)
            val = li.sum()                                     # (the kernel's items carry the weight)
            if not want_grad:
                return val
            gH = ops.irfft_pow2_bwd(gx_sh.view(B * C, n), n, window=(start, start + L)).reshape(B, C, K)
            return _ScalarLossWithSavedGrad.apply(H_sh, val, gH, unit_grad)
        # pairs of real samples as complex numbers: the streaming mix kernel of the spectra serves the signals too
        x_dir = ops.sh_to_directional(analysis_matrix, torch.view_as_complex(x_sh.view(B, C, n // 2, 2)), False)
        x_dir = torch.view_as_real(x_dir).view(B * J, n)
        maskw, count = self._time_mask(L, x_dir.device)
        li, gx = ops.edc_loss_model(x_dir, start, L, amps, self.envelopes, maskw, 1.0 / (B * J * count), float(weight),
                                    want_grad)
        val = li.sum() if weight == 1.0 else li.sum() * float(weight)
        if not want_grad:
            return val
        gx_sh = ops.sh_to_directional(analysis_matrix, torch.view_as_complex(gx.view(B, J, n // 2, 2)), True)
        gH = ops.irfft_pow2_bwd(torch.view_as_real(gx_sh).view(B * C, n), n).reshape(B, C, K)
        return _ScalarLossWithSavedGrad.apply(H_sh, val, gH, unit_grad)

    def lines_supported(self, K: int, G: int, nper: int, J: int, S: int) -> bool:
        """Shapes forward_lines takes (otherwise: SHOutputStage + forward_sh)."""
        return 2 * (K - 1) == 131072 and ops.dirlin_supported(G, nper) and ops.edc_mixed_supported(nper, J, S)

    def forward_lines(self, Y: torch.Tensor, c: torch.Tensor, w: torch.Tensor, G: int, nper: int,
                      filt: Optional[torch.Tensor], analysis_matrix: torch.Tensor, amps_true: torch.Tensor,
                      weight: float = 1.0, unit_grad: bool = False) -> torch.Tensor:
        """``forward_sh(SHOutputStage(Y, c, w, filt), A, amps_true)`` (reference model.py:1056-1088, trainer.py:853-865,
        losses.py:333-371) with the output stage behind the inverse transform as well: H_sh[b][l] is linear in the receiver's
        SH weights, irfft(H_sh[b][l]) = sum_g w[b][g nper + l] irfft(c_n filt Y[:, n]) -- the N = G nper line responses are
        transformed (27 instead of 288 transforms at order 2, three groups, 32 receivers), the receivers' SH signals are
        formed on the EDC window only, and the adjoint runs the same way back (csrc/dirlin.hip).  Y (K, N) complex: the
        delay-line responses of the transposed solve; c (N,) output gains; w (B, N) SH weights; A (J, nper)."""
        K, N = Y.shape
        B = w.shape[0]
        J = analysis_matrix.shape[0]
        n = 2 * (K - 1)
        start = self.mixing_time_samps
        L = min(self.edc_len_samps, n - start)
        if L <= 0:
            raise ValueError("EDC window starts beyond the impulse-response length")
        want_grad = (Y.requires_grad or c.requires_grad or w.requires_grad) and torch.is_grad_enabled()
        tau = ops.irfft_pow2_fwd(ops.dirlin_lines_fwd(Y, c, filt), n)                   # (N, n)
        if self.envelopes.device != tau.device:
            self.envelopes = self.envelopes.to(tau.device)
        amps = amps_true.to(device=tau.device, dtype=torch.float32).reshape(B * J, -1).contiguous()
        wf = w.detach().reshape(B, N)                          # ((B, G, nper) from the weights network)
        x_sh = ops.dirlin_combine(tau, start, L, wf, G, nper)                           # (B, nper, ceil4(L)): the window
        maskw, count = self._time_mask(L, tau.device)
        li, gx_sh = ops.edc_loss_model_mixed(x_sh, analysis_matrix, 0, L, amps, self.envelopes, maskw,
                                             1.0 / (B * J * count), float(weight), want_grad)
        val = li.sum()                                         # (the kernel's items carry the weight)
        if not want_grad:
            return val
        gtau, gw = ops.dirlin_gamma_dots(gx_sh, L, tau, start, wf, G, nper)
        gZ = ops.irfft_pow2_bwd(gtau, n, window=(start, start + L))
        gY, gc = ops.dirlin_lines_bwd(Y, c, filt, gZ)
        return _ScalarLossWithSavedGrads.apply(val, unit_grad, Y, c, w, gY, gc, gw)
