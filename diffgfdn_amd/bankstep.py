"""The band bank's optimiser step as an explicit launch sequence (no autograd), on the polynomial form of the
block transfer functions (csrc/blocktf.hip).

What the reference does per batch and band (src/diff_gfdn/trainer.py:373-379, :452-477): ``normalize`` (no-grad
sub-FDN forward, b, c /= E^(1/4)), forward (model.py:569-625), losses (trainer.py:259-315), ``backward()``,
``optimizer.step()``.  With zero coupling and blocks of at most four delay lines every one of those stages sees the
feedback loop only through the group transfer functions T_g(z) -- ratios of multilinear polynomials in the phasors
z^{m_i} with 2 x 16 real coefficients per block -- so the step is

    side  : records of the raw blocks M_g -> energy pass (normalize: b, c rescaled in place, scale_g)
            -> [mask draw] -> colorless pass (spectral loss + dL/drecords) -> sparsity gradient
    side2 : gain network forward ................................ EDC scans ........ gain network backward
    main  : Q, QQ = expm -> records of Q_g Q_g (at the rescaled b, c) -> output stage H -> irfft -> STFT / EDR
            -> STFT adjoint -> irfft adjoint -> output-stage adjoint (dL/drecords, dL/dgains)
            -> records -> (dL/dQQ, dL/dM_raw, dL/db, dL/dc) -> expm adjoint -> [all-reduce] -> Adam

26 launches per step of all bands; every gradient lands directly in the optimiser's flat gradient buffer (no
accumulate / pack kernels), the (K, N) delay-line responses of the per-bin solve never exist.  The autograd
path of ``BandBankTrainer._step_losses`` (per-bin elimination kernels) stays as the general fallback and as the
cross-check of this one (tests/test_gpu_bank.py).
"""
from typing import Dict, Optional

import torch

from . import hip_ops as ops
from .functional import FrequencyGrid


class FusedBankStep:
    """Explicit forward / backward of ``BandBankTrainer`` for blocks of <= 4 lines, <= 4 groups, <= 64 receivers
    per band.  ``supported(trainer)`` tells whether a trainer's layout qualifies."""

    @staticmethod
    def supported(trainer) -> bool:
        bank = trainer.net
        return (bank.num_delay_lines_per_group <= 4 and bank.num_groups <= 4
                and bank.num_bands * bank.num_groups <= 64)

    def __init__(self, trainer):
        self.tr = trainer
        bank, opt = trainer.net, trainer.optimizer
        views = {id(p): v for p, v in zip(opt._params, opt._grad_views)}
        # the flat gradient buffer's slices of the four stacked leaves: kernels write them directly
        self.g_c = views[id(bank.output_gains)].view(-1)
        self.g_b = views[id(bank.input_gains)].view(-1)
        self.g_w = views[id(bank.output_scalars_w)].view(-1)
        self.g_M = views[id(bank.feedback_loop_M)].view(bank.num_bands * bank.num_groups,
                                                        bank.num_delay_lines_per_group,
                                                        bank.num_delay_lines_per_group)
        dev = bank.input_gains.device
        nblk = bank.num_bands * bank.num_groups
        self._zero_loss = torch.zeros(nblk, dtype=torch.float32, device=dev)
        self._keep = []

    # ------------------------------------------------------------------------------------------
    def _decay_middle(self, H, K, rows, maskw, inv, want_grad, order, edr_t, edc_t, start, length):
        """irfft -> EDR / EDC losses -> dL/dH (slot order / pair-interleaved when the length allows it).
        Returns (li_edr partials, li_edc, gH or None, sums = per-band [total, w_edr edr, w_edc edc])."""
        tr, cfg = self.tr, self.tr.config
        main = torch.cuda.current_stream()
        side2 = tr._stream('_side2')
        B = H.shape[0]
        win = tr.stft_win
        pairs = order is not None and tr.use_pairs and win == 4096
        keep = self._keep
        if pairs:
            x = ops.irfft_odd_fwd(H, K, slots=True, pairs=True)
        else:
            x = ops.irfft_odd_fwd(H, K, slots=order is not None)
        keep.append(x)
        if side2 is not None:
            side2.wait_stream(main)
        with torch.cuda.stream(side2) if side2 is not None else _null():
            if pairs:
                li_edc, g_edc = ops.edc_loss_pairs(x, B, start, length, edc_t, maskw, inv, cfg.edc_loss_weight,
                                                   want_grad, rows=rows)
            else:
                li_edc, g_edc = ops.edc_loss(x, start, length, edc_t, maskw, inv, cfg.edc_loss_weight, want_grad,
                                             rows=rows)
            keep.extend((li_edc, g_edc))
        T_edr, sum_abs = edr_t
        if pairs:
            P = ops.stft_power_pairs(x, B, win)
            g_edr = None
        else:
            g_edr = torch.empty_like(x) if want_grad else None
            P = ops.stft_power(x, win, zero_buf=g_edr)
        li_edr = ops.edr_loss(P, T_edr, sum_abs, None, cfg.edr_loss_weight, want_grad, rows=rows, defer=True)
        keep.extend((P, li_edr, g_edr))
        if side2 is not None:
            main.wait_stream(side2)
        # the reported sums ride the EDC stream beside the adjoint kernels
        if side2 is not None:
            side2.wait_stream(main)
        with torch.cuda.stream(side2) if side2 is not None else _null():
            sums = ops.weighted_sums(li_edr, cfg.edr_loss_weight, li_edc, cfg.edc_loss_weight, sum_abs, rows,
                                     tr.num_bands)
            self._ev_sums = torch.cuda.Event()
            self._ev_sums.record()
        gH = None
        if want_grad:
            if pairs:
                g = ops.stft_power_pairs_bwd(x, B, win, P, base=g_edc, out=g_edc)
                gH = ops.irfft_odd_pairs_bwd(g, K, B)
            else:
                g_edr = ops.stft_power_bwd(x, win, P, g_edr)
                gH = ops.irfft_odd_bwd(g_edc, K, H.shape[1], g_edr, slots=order is not None)
            keep.append(gH)
        return li_edr, li_edc, gH, sums

    @torch.no_grad()
    def run(self, data: Dict, maskw: Optional[torch.Tensor], inv: float, normalize_first: bool, train: bool,
            allreduce=None, opt_step: bool = True, mask_draw=None) -> Dict:
        """One step of every band on the band-major batch ``data`` (collate(lean="rows")).  ``maskw``: EDC time
        weights (None: no mask), ``inv``: what the EDC terms are divided by beyond the weights (1 when the weights are
        pre-normalised).  ``train``: gradients into the flat buffer, [all-reduce,] Adam (``opt_step=False`` stops in front
        of the all-reduce: the caller runs it and the update).  ``mask_draw``: callable that fills ``maskw`` on the
        device (run on the side stream, off the path to the output stage).  Returns the loss dict of
        ``BandBankTrainer._step_losses`` (+ '_total')."""
        tr = self.tr
        bank, cfg, nb = tr.net, tr.config, tr.num_bands
        G, n = bank.num_groups, bank.num_delay_lines_per_group
        z, rows = data['z_values'], data['row_index']
        Btot = rows.numel()
        if Btot % nb or Btot // nb > 64:
            raise ValueError("the batch must hold the same number (<= 64) of receivers for every band")
        K = z.shape[-1]
        keep = self._keep
        main = torch.cuda.current_stream()
        side, side2 = tr._stream('_side'), tr._stream('_side2')
        for s_ in (side, side2):
            if s_ is not None:
                s_.wait_stream(main)
        M = bank._blocks().detach()
        b, c = bank.input_gains.data.view(-1), bank.output_gains.data.view(-1)
        delays, ig = bank.delays, bank.inv_gamma
        gridK = FrequencyGrid.of(z)
        order = ops.irfft_slot_order(K, z.device) if (tr.use_slot_order and 'dataset' in data) else None
        Ku = (K + 1) // 2 if K % 2 == 1 else K
        if order is not None:
            zu, direct = data['dataset'].slot_ordered(*order)
        else:
            zu, direct = z[:Ku], data['target_early_response'][:, :Ku]
        gridU = FrequencyGrid.of(zu)
        filt = tr._filter_on(Ku, order)
        inv_world = 1.0 / tr.world_size
        ev_norm, ev_mlp, ev_side = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()

        # side: records of the raw blocks -> normalize (trainer.py:317-332)
        with torch.cuda.stream(side) if side is not None else _null():
            coef_sub = ops.tf_coefs(M, b, c, None)
            scale = None
            if normalize_first:
                _, scale = ops.tf_energy(gridK.turns, gridK.logr, coef_sub, delays, n, b, c, want_energy=False)
            ev_norm.record()
            keep.extend((coef_sub, scale))
        # side2: receiver gains (gain_filters.py:497-536)
        Hh, n_hidden, _, lo, hi = bank._mlp_cfg
        w = bank.output_scalars_w.detach()
        with torch.cuda.stream(side2) if side2 is not None else _null():
            rgain, xhat, rstd = ops.mlp_gains_fwd(data['norm_listener_position'], bank._freq_pi, w, Hh, n_hidden, G,
                                                  lo, hi, rows, nb)
            ev_mlp.record()
            keep.extend((rgain, xhat, rstd))
            # the EDC time mask is drawn on the stream that runs the EDC scans.  (Drawn on `side`, with the reported
            # total on `side` waiting for the sums of `side2`, the two forked streams depend on each other in both
            # directions -- hipStreamEndCapture of ROCm 7.2 segfaults on that topology.)
            if mask_draw is not None:
                mask_draw()
        # main: rotations, records of the damped loop at the rescaled gains, output stage
        Q, QQ = ops.ortho_fwd(M, True, True)
        main.wait_event(ev_norm)
        coef = ops.tf_coefs(QQ, b, c, ig)
        main.wait_event(ev_mlp)
        H = ops.tf_compose_fwd(gridU.turns, gridU.logr, coef, delays, n, rgain, None, direct, filt, rows, nb)
        keep.extend((Q, QQ, coef, H))
        # side: colorless pass + sparsity gradient + the reported colorless terms (issued behind the
        # output stage: the graph executor launches nodes in capture order)
        if side is not None:
            side.wait_stream(main)             # Q
        with torch.cuda.stream(side) if side is not None else _null():
            grec_sub, loss_g = ops.tf_colorless(gridK.turns, gridK.logr, coef_sub, delays, n, scale,
                                                cfg.use_asym_spectral_loss, cfg.spectral_loss_weight * inv_world)
            out3, gQ = ops.colorless_terms(loss_g, Q, cfg.spectral_loss_weight, cfg.sparsity_loss_weight,
                                           inv_world, want_grad=train, nbands=nb)
            ev_side.record()
            keep.extend((grec_sub, loss_g, gQ))
        start, length = tr._decay_window(K)
        edr_t, edc_t = data['edr_target'], data['edc_target']
        li_edr, li_edc, gH, sums = self._decay_middle(H, K, rows, maskw, inv, train, order, (edr_t[1], edr_t[2]),
                                                      edc_t[1], start, length)
        with torch.cuda.stream(side) if side is not None else _null():       # the reported total, beside the adjoints
            torch.cuda.current_stream().wait_event(self._ev_sums)
            total = (sums[:, 0] + out3[:, 0]) if nb > 1 else (sums[0] + out3[0])
            keep.extend((sums, out3))
        if train:
            grec, grg = ops.tf_compose_bwd(gridU.turns, gridU.logr, coef, delays, n, rgain, gH, None, filt, nb)
            keep.extend((grec, grg))
            ev_cb, ev_mlpb = torch.cuda.Event(), torch.cuda.Event()
            ev_cb.record()
            if side2 is not None:
                side2.wait_event(ev_cb)
            with torch.cuda.stream(side2) if side2 is not None else _null():
                ops.mlp_gains_bwd(data['norm_listener_position'], bank._freq_pi, w, Hh, n_hidden, G, lo, hi, rgain,
                                  xhat, rstd, grg, rows, nb, out=self.g_w)
                ev_mlpb.record()
            main.wait_event(ev_side)
            gQQ, gMsub, _, _ = ops.tf_coefs_bwd(QQ, ig, grec, b, c, A1=M, grec1=grec_sub, gb=self.g_b, gc=self.g_c)
            ops.ortho_bwd_add(M, gQ, gQQ, Q, gMsub, out=self.g_M)
            keep.extend((gQQ, gMsub))
            main.wait_event(ev_mlpb)
            tr.optimizer._packed = True               # the flat gradient buffer is complete
            if opt_step:
                if allreduce is not None:
                    allreduce()
                tr.optimizer.step()
        else:
            main.wait_event(ev_side)
        if nb > 1:
            losses = {'edc_loss': sums[:, 2], 'edr_loss': sums[:, 1], 'spectral_loss': out3[:, 1],
                      'sparsity_loss': out3[:, 2], '_total': total}
        else:
            losses = {'edc_loss': sums[2], 'edr_loss': sums[1], 'spectral_loss': out3[1], 'sparsity_loss': out3[2],
                      '_total': total}
        for s_ in (side, side2):
            if s_ is not None:
                main.wait_stream(s_)
        keep.clear()          # every consumer is ordered before the next step's first launch on each stream
        return losses


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
