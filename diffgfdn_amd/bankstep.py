"""The band bank's optimiser step as an explicit launch sequence (no autograd), on the polynomial form of the
block transfer functions (csrc/blocktf.hip).

What the reference does per batch and band (src/diff_gfdn/trainer.py:373-379, :452-477): ``normalize`` (no-grad
sub-FDN forward, b, c /= E^(1/4)), forward (model.py:569-625), losses (trainer.py:259-315), ``backward()``,
``optimizer.step()``.  With zero coupling and blocks of at most eight delay lines every one of those stages sees the
feedback loop only through the group transfer functions T_g(z) -- ratios of multilinear polynomials in the phasors
z^{m_i} with 2 x 16 (<= 4 lines) or 2 x 256 (5..8 lines, evaluated on the matrix cores) real coefficients per block --
so the step is (the timed form: the linear step with all of round 5's arrangements; DESIGN.md section 5 has the launch
sequence with times, the class attributes below select the older forms for cross-checks)

    main  : group responses T_g filt of the band's G groups from the PERSISTENT records (left by the previous step's tail)
            -> their inverse transform (3 passes; unscaled: normalize's scale sits in the receiver gains) -> their STFT
            -> [gains] EDR loss on spectra composed per receiver as Sd[row] + sum_g gain[b][g] STFT(tau_g), with the sums of
            the gradient spectra over the band's receivers in the same launch -> adjoint STFT -> merge with the EDC part
            -> adjoint transform (3 passes) -> records pass -> tail: parameter gradients -> Adam on M, b, c -> the NEXT
            step's Q, Q Q and record sets (one launch)
    side2 : energy pass (normalize) -> gain network forward -> finish (b, c rescaled, scale, gains x scale) -> mask draw
            -> [group signals] EDC term (one register-resident launch per receiver) -> sum of dL/dx over the band's
            receivers -> colorless pass + sparsity terms -> reported sums -> gain network backward -> its Adam range
            -> next step's receivers

Blocks of 5..8 lines: the same with the coefficient records (csrc/blocktf8.hip) in place of the persistent records, the
forward group responses on the matrix cores and normalize / colorless pass / both adjoints as real transforms of the
coefficient sequences (csrc/polyfft.hip) where the delay lengths are integers on the reference's own grid.

28 launches per step of all bands (4-line blocks); every gradient lands directly in the optimiser's flat gradient buffer (no
accumulate / pack kernels), neither the (K, N) delay-line responses of the per-bin solve nor any per-receiver spectrum or
time signal exists.  The autograd path of ``BandBankTrainer._step_losses`` (per-bin elimination kernels) stays as the
general fallback and as the cross-check of this one (tests/test_gpu_bank.py).
"""
from typing import Dict, Optional

import torch

from . import hip_ops as ops
from .functional import FrequencyGrid
from .losses import shard_loss_scales


class FusedBankStep:
    """Explicit forward / backward of ``BandBankTrainer`` for zero-coupling blocks of <= 8 lines and <= 4 groups per band
    (<= 4 lines: csrc/blocktf.hip; 5..8 lines: csrc/blocktf8.hip).  ``supported(trainer)`` tells whether a trainer's
    layout qualifies."""

    @staticmethod
    def supported(trainer) -> bool:
        """Blocks of <= 4 lines: any grid; blocks of 5..8 lines: grids on the unit circle only (``run`` refuses others, and
        the trainer's configuration is the only place that knows before the first batch arrives)."""
        bank = trainer.net
        if bank.num_delay_lines_per_group > 4 and getattr(trainer.config, 'reduced_pole_radius', 1.0) != 1.0:
            return False
        return (bank.num_delay_lines_per_group <= 8 and bank.num_groups <= 4
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
        self._keep_prev = []
        self.w_range = opt.flat_range(bank.output_scalars_w)
        if self.w_range[1] != opt.flat_grad.numel():
            raise RuntimeError("the gain network's parameters must close the flat buffers (BandBankTrainer's group order)")
        self._off = (opt.flat_range(bank.feedback_loop_M)[0], opt.flat_range(bank.input_gains)[0],
                     opt.flat_range(bank.output_gains)[0])
        # Q, QQ and the two record sets of the CURRENT parameters (blocks of <= 4 lines): the fused tail of a training step
        # leaves the next step's here, so that a step does not start with their launch
        self._rec = None
        self._rec_valid = False
        self._rec_versions = None

    # -- records that outlive a step -------------------------------------------------------------------------------------
    def _records(self):
        if self._rec is None:
            bank = self.tr.net
            nblk, n = bank.num_bands * bank.num_groups, bank.num_delay_lines_per_group
            dev = bank.input_gains.device
            mk = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            if n > 4:
                # blocks of 5..8 lines: Q, Q Q and the snapshot of the output gains the forward pass reads (the coefficient
                # records themselves are a launch of 500 workgroups at the step's head: csrc/blocktf8.hip)
                self._rec = (mk(nblk, n, n), mk(nblk, n, n), mk(nblk * n))
            else:
                self._rec = (mk(nblk, n, n), mk(nblk, n, n), mk(nblk, 32), mk(nblk, 32))
        return self._rec

    def _versions(self):
        bank = self.tr.net
        leaves = [bank.feedback_loop_M, bank.input_gains, bank.output_gains]
        for net in getattr(bank, 'nets', ()):              # (the bands' own modules hold views of the stacked leaves)
            leaves.extend(p for name, p in net.named_parameters() if 'output_scalars' not in name)
        return tuple(p._version for p in leaves)

    def records_ok(self) -> bool:
        """Whether the kept records belong to the current parameters (host-side bookkeeping: every path of this package that
        changes M, b, c calls :meth:`invalidate_records` or bumps a tensor version)"""
        return self._rec_valid and self._rec_versions == self._versions()

    def invalidate_records(self):
        self._rec_valid = False

    @torch.no_grad()
    def prime_records(self):
        """Records of the current parameters into the kept buffers (one launch on the current stream)"""
        bank = self.tr.net
        if bank.num_delay_lines_per_group > 4:
            Q, QQ, cs = self._records()
            q, qq = ops.ortho_fwd(bank._blocks().detach(), True, True)
            Q.copy_(q)
            QQ.copy_(qq)
            cs.copy_(bank.output_gains.data.view(-1))
            self._rec_valid, self._rec_versions = True, self._versions()
            return
        ops.tf_ortho_coefs(bank._blocks().detach(), bank.inv_gamma, bank.input_gains.data.view(-1),
                           bank.output_gains.data.view(-1), out=self._records())
        self._rec_valid, self._rec_versions = True, self._versions()

    def ensure_records(self):
        """In front of a graph replay whose capture found valid records (and therefore holds no records launch)"""
        if self._rec is not None and not self.records_ok():
            self.prime_records()

    # ---- forms of the step.  Plain class attributes: the defaults are the timed path; tests switch single ones off to check a
    # form against the one it replaced (tests/test_gpu_bank.py, tests/test_gpu_fullsize.py).  What was measured and did not
    # pay is described in DESIGN.md section 4.3, not kept here.
    #
    # The output stage in the TIME domain (csrc/linear.hip): the inverse transform is linear and neither the group transfer
    # functions nor the band's filter depend on the receiver, so x[b] = xd[row_b] + sum_g rgain[b][g] irfft(s_g T_g filt) with
    # xd = irfft(direct filt) a constant of the dataset (BandStackedDataset.direct_time).  A step then runs G forward and G
    # adjoint transforms per band instead of one per receiver (28 instead of 224 at the north-star size), dL/drgain is a
    # dot product over the time samples and the records pass reads G adjoint spectra per band.  Off: the folded output
    # stage below (round 3's path; also what receiver counts outside the linear kernels' limits and chained steps take).
    linear_transforms = True
    # ... and the EDR loss on linearly COMPOSED short-time spectra (csrc/edrlin.hip): the STFT is linear too, so a receiver's
    # spectrum is Sd[row] + sum_g rgain[b][g] STFT(tau_g) with Sd a constant of the dataset (BandStackedDataset.direct_stft).
    # A step runs G forward and G adjoint STFTs per band; per receiver only streaming arithmetic on (frame, frequency) cells
    # is left and the receivers' signals are never stored.
    spectral_edr = True
    # ... with the sum of the gradient spectra over the band's receivers inside the EDR launch (k_edr_lin_wave: a thread owns
    # cells of the band's plane and walks the receivers; dL/d|S|^2 never exists, the direct-path spectra are read once), the
    # band's receivers cut into runs for more loads in flight.  Off: the value-only column kernel + the sum as two launches
    # (the cross-check form)
    edr_one_launch = True
    # (0 = two runs: measured same-box, 7 bands: 1 / 2 / 4 runs 0.379 / 0.358 / 0.370 ms; one band: 2 / 4 / 8 / 16 runs 0.222 /
    # 0.227 / 0.229 / 0.229 ms -- every run is one more set of partial planes for the adjoint STFT to add on load)
    edr_receiver_runs = 0

    def _edr_runs(self, nbands: int, B: int) -> int:
        if self.edr_receiver_runs > 0:
            return min(self.edr_receiver_runs, B)
        return 2 if B >= 4 else 1
    # ... on planes stored in the tiled cell order (frequency blocks of 256, a block's frames contiguous): what a (receiver,
    # frequency block) workgroup of the EDR kernel touches is one contiguous run
    tiled_spectra = True
    # ... and the G sums of the EDC gradient signals together with the EDC part of dL/drgain in one sweep over the window
    gamma_dots_one_launch = True
    # ... and the adjoint STFT of the gradient spectra as ONE launch for all frames (odd frames to a second signal set that the
    # gamma sweep adds) instead of the even-frame launch with the odd-frame launch behind it
    adjoint_stft_one_launch = True

    # Round 5.  (a) The EDC term as ONE register-resident launch per receiver (csrc/edcone.hip: compose, both scans, dB stage,
    # dL/dx and the EDC part of dL/drgain without staging the window samples or dL/dEDC through memory) and the light gamma
    # sweep behind it (no dot products left to do) in place of the three scan launches + gfdn_lin_gamma_dots (which stay
    # for windows beyond the launch's 49 152 samples).
    edc_one_launch = True
    # Round 6: the gain network in front of the energy pass on the side stream (None: only for banks whose bands have their
    # own network sizes, where the network's launch is the long one)
    network_first = None
    # (b) normalize OFF the critical chain: T is bilinear in b, c and the transform is linear, so the normalisation scale can
    # join the group signals where the forward transform's LAST pass stores them.  The energy pass then runs on the side
    # stream beside the group responses and the transform's first two passes instead of in front of them (~28 us of chain).
    scale_late = True
    # ... and INSIDE the receiver gains (round 5, late in the round): H = sum_g gain[b][g] (s_g T_g) + direct is linear in
    # both factors, so the finish launch of the energy pass stores gain s beside the gains (the gain network's forward runs in
    # front of the energy pass on the side stream), every launch of the linear step takes those as the receiver gains on group
    # signals of the UNSCALED functions, the network's backward multiplies its incoming rows by s and the records pass divides
    # dL/dT by s.  The transform's last pass then waits for nothing (it waited ~18 us for the scale: the side stream's start-up
    # inside a graph is ~20 us, the energy pass 26 -- later than the transform's first two passes take).
    scale_in_gains = True
    # (c) tail and head as one launch (single process): records -> parameter gradients -> Adam on the blocks' own M, b, c
    # -> the NEXT step's Q, QQ and record sets (csrc/blocktf.hip k_tf_tail); the gain network's range of the flat buffers is
    # stepped on the side stream behind its own backward.  A step then starts with the group responses.
    fused_tail = True
    # (d) the EDC part of dL/dtau summed over the band's receivers on the side stream, straight behind the EDC launch; the
    # main chain merges it with the adjoint STFT's two signal sets (three small arrays) into the transform's slot order
    gamma_split = True
    # (e) blocks of 5..8 lines: the polynomial passes as real transforms of the coefficient sequences (csrc/polyfft.hip) where
    # the delay lengths are integers and the grid is the reference's rfftfreq grid -- ~45 transform-flops per bin and
    # polynomial instead of 512 products; the matrix-core passes (csrc/blocktf8.hip) stay for every other grid
    transform_polys = True

    # The output stage H = (sum_g rgain s_g T_g + direct) filt formed INSIDE the first pass of the forward transform
    # (gfdn_irfft_odd_pairs_compose_fwd) from the saved group transfer functions: H is neither written nor read back
    # (the stored-signal path, linear_transforms = False).
    fold_output_stage = True
    # The mirror image -- the gains pass of the adjoint inside the LAST pass of the adjoint transform
    # (gfdn_irfft_odd_pairs_gains_bwd: dL/dH used while it is in registers) -- is OFF: measured 0.676 against 0.664 ms.
    fold_gains = False
    gain_rows_in_mlp = True
    colorless_behind_scans = True      # (blocks of 5..8 lines: the colorless pass behind the EDC launches)
    colorless_late_small = True        # (blocks of <= 4 lines on the spectral-EDR step: behind them as well)
    # the optimiser update on the side stream behind the gain network's backward (single process)
    adam_on_side = True

    def _eye_rows(self, nb: int, G: int, device) -> torch.Tensor:
        """(nb * G, G): "receiver" g of every band with gain 1 on group g -- what turns the output-stage kernels into
        H_g = T_g filt and its adjoint (cached: captured graphs hold its pointer)"""
        key = (nb, G, str(device))
        if getattr(self, '_eye_cache', None) is None or self._eye_cache[0] != key:
            self._eye_cache = (key, torch.eye(G, dtype=torch.float32, device=device).repeat(nb, 1).contiguous())
        return self._eye_cache[1]

    # ------------------------------------------------------------------------------------------
    def _decay_middle(self, H, K, rows, maskw, inv, train, order, pairs, T_edr, sum_abs, T_edc, start, length, ev, main,
                      side2, x_fn=None, Btot=None, gains=None, item_len=None, Bper=None, lin=False):
        """One chain over the whole batch: main: irfft -> STFT -> EDR -> STFT adjoint (even frames, then odd frames +
        EDC gradient) -> irfft adjoint; side2: EDC scans.  Records ev['x'] / ev['edc'] / ev['g'].
        ``lin``: the time-domain output stage -- the chain stops at dL/dx, returned as (g, None) (pair-interleaved, merged)
        or (g_edc, g_edr) in place of dL/dH."""
        tr, cfg, keep = self.tr, self.tr.config, self._keep
        Btot, win = (H.shape[0] if Btot is None else Btot), tr.stft_win
        on_side2 = (lambda: torch.cuda.stream(side2)) if side2 is not None else _null
        P = None
        if x_fn is not None:              # (the output stage rides the transform's first pass: H is never stored)
            x = x_fn()
            if isinstance(x, tuple):      # (... or the STFT's load, which then also returns |STFT|^2)
                x, P = x
        elif pairs:
            x = ops.irfft_odd_fwd(H, K, slots=True, pairs=True)
        else:
            x = ops.irfft_odd_fwd(H, K, slots=order is not None)
        ev['x'].record()
        if pairs:
            if P is None:
                P = ops.stft_power_pairs(x, Btot, win)
            g_edr = None
        else:
            g_edr = torch.empty_like(x) if train else None
            P = ops.stft_power(x, win, zero_buf=g_edr)
        with on_side2():
            torch.cuda.current_stream().wait_event(ev['x'])
            if pairs:
                li_edc, g_edc = ops.edc_loss_pairs(x, Btot, start, length, T_edc, maskw, inv, cfg.edc_loss_weight,
                                                   train, rows=rows, item_len=item_len, items_per_band=Bper)
            else:
                li_edc, g_edc = ops.edc_loss(x, start, length, T_edc, maskw, inv, cfg.edc_loss_weight, train,
                                             rows=rows, item_len=item_len, items_per_band=Bper)
            ev['edc'].record()
        li_edr = ops.edr_loss(P, T_edr, sum_abs, None, cfg.edr_loss_weight, train, rows=rows, defer=True)
        keep.extend((x, P, g_edr, li_edc, g_edc, li_edr))
        gH = None
        if train:
            if pairs:
                # even frames first, alone; the EDC gradient joins with the odd frames (the EDC scans are the
                # longer of the two branches: the main stream would otherwise sit idle until they finish)
                g = ops.stft_power_pairs_bwd(x, Btot, win, P, phase=0)
                main.wait_event(ev['edc'])
                ops.stft_power_pairs_bwd(x, Btot, win, P, base=g_edc, out=g, phase=1)
                ev['g'].record()
                if lin:
                    gH = (g, None)
                elif gains is not None:     # (the gains pass of the output stage's adjoint rides the last pass)
                    gH, self._gpart = ops.irfft_odd_pairs_bwd(g, K, Btot, gains=gains)
                    keep.append(self._gpart)
                else:
                    gH = ops.irfft_odd_pairs_bwd(g, K, Btot)
                keep.append(g)
            else:
                g_edr = ops.stft_power_bwd(x, win, P, g_edr)
                main.wait_event(ev['edc'])
                ev['g'].record()
                if lin:
                    gH = (g_edc, g_edr)
                else:
                    gH = ops.irfft_odd_bwd(g_edc, K, (K + 1) // 2 if order is not None else H.shape[1], g_edr,
                                           slots=order is not None)
            keep.append(gH)
        else:
            main.wait_event(ev['edc'])
            ev['g'].record()
        return li_edr, li_edc, gH

    def _decay_middle_spec(self, data, K, rows, rgain, tau, xd, maskw, inv, train, T_edr, sum_abs, T_edc, start, length,
                           item_len, nb, G, ev, main, side2, wait_gains):
        """The decay losses without per-receiver transforms of any kind (csrc/edrlin.hip, xlin_dev.h):
            main : STFT of the band's G group signals -> EDR loss on composed spectra (loss partials, dL/d|S|^2, EDR part of
                   dL/drgain) -> gradient spectra summed over the band's receivers -> their adjoint STFT (EDR part of dL/dtau)
            side2: EDC scans on samples formed where they are read (-> dL/dx of the EDC term, pair-interleaved)
        Returns (li_edr partials, li_edc, (g_edc, gam_edr, parts) or None); records ev['x'] / ev['edc'] / ev['g']."""
        tr, cfg, keep = self.tr, self.tr.config, self._keep
        Btot, win = rows.numel(), tr.stft_win
        ds = data['dataset']
        tiled = self.tiled_spectra
        Sd = ds.direct_stft(tr.subband_filter_freq_resp, K, win, tiled=tiled)
        if tiled:
            T_edr = ds.edr_target_tiled()
        on_side2 = (lambda: torch.cuda.stream(side2)) if side2 is not None else _null
        ev['x'].record()                                     # (tau complete)
        edc_one = self._edc_one(length, G)
        nch = 1 if edc_one else (ops.lin_gamma_dots_tiles(K) if self.gamma_dots_one_launch else ops.lin_gain_chunks(K))
        parts = None
        if train:
            # partial rows of dL/drgain: [EDC dot products | EDR columns]; the gain network's backward sums them
            parts = torch.empty((Btot * G, nch + ops.edr_lin_parts(win // 2 + 1, fused=self.edr_one_launch)),
                                dtype=torch.float32, device=rows.device)
        Stau = ops.stft_pairs_spectrum(tau, nb * G, win, tiled=tiled)
        with on_side2():
            torch.cuda.current_stream().wait_event(ev['x'])
            if edc_one:
                # ONE launch, one workgroup per receiver: loss, dL/dx on the window (plain rows) and the EDC column of
                # ``parts``
                li_edc, g_edc = ops.edc_lin_one(xd, rows, tau, rgain, nb, K, start, length, T_edc, maskw, inv,
                                                cfg.edc_loss_weight, train, trows=rows, item_len=item_len, dots=parts,
                                                col=0)
            else:
                li_edc, g_edc = ops.edc_loss_pairs_lin(xd, rows, tau, rgain, nb, K, start, length, T_edc, maskw, inv,
                                                       cfg.edc_loss_weight, train, trows=rows, item_len=item_len,
                                                       fill_outside=not self.gamma_dots_one_launch)
            if edc_one and train and self.gamma_split:
                # the EDC part of dL/dtau right here, beside the EDR launch: the main chain then merges three small signal
                # sets behind the adjoint STFT instead of sweeping the receivers' gradient rows (42 MB) there
                _, band_len = tr._item_windows(K, Btot // nb, rows.device)
                g_edc = (g_edc, ops.lin_gamma_win(g_edc, rgain, nb, K, start, length, band_win_len=band_len))
            ev['edc'].record()
        wait_gains()
        gP = Gs = None
        if train and self.edr_one_launch:
            li_edr, Gs = ops.edr_lin_loss_gsum(Sd, rows, Stau, rgain, nb, T_edr, sum_abs, cfg.edr_loss_weight, dots=parts,
                                               col0=nch, tiled=tiled, nsplit=self._edr_runs(nb, Btot // nb))
        else:
            li_edr, gP = ops.edr_lin_loss(Sd, rows, Stau, rgain, nb, T_edr, sum_abs, cfg.edr_loss_weight, train,
                                          dots=parts, col0=nch, tiled=tiled)
        if not (train and self.gamma_dots_one_launch):
            # (training: the side stream's consumers of the EDR partials wait for the LATER event behind the gamma merge --
            # one edge from the main chain to the side stream instead of two; an edge costs the main chain ~4 us)
            ev['g'].record()                                 # (EDR partials and the EDR columns of ``parts`` complete)
        keep.extend((Sd, Stau, li_edc, g_edc, li_edr, gP, parts))
        if not train:
            main.wait_event(ev['edc'])
            return li_edr, li_edc, None
        if Gs is None:
            Gs = ops.edr_lin_gsum(Sd, rows, Stau, rgain, nb, gP)
        # (one launch for all frames: the odd frames' contributions go to a second signal set that the gamma sweep adds)
        one = self.adjoint_stft_one_launch and self.gamma_dots_one_launch
        gam_edr = ops.stft_pairs_spectrum_bwd(Gs, K, nb * G, win, tiled=tiled, split_parity=one)
        keep.extend((Gs, gam_edr))
        return li_edr, li_edc, (g_edc, gam_edr, parts)

    def _edc_one(self, length: int, G: int) -> bool:
        return self.edc_one_launch and self.gamma_dots_one_launch and ops.edc_lin_one_supported(length, G)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def run(self, data: Dict, maskw: Optional[torch.Tensor], inv: float, normalize_first: bool, train: bool,
            allreduce=None, opt_step: bool = True, mask_draw=None, tail=None, pipe=None) -> Dict:
        """One step of every band on the band-major batch ``data`` (collate(lean="rows")).  ``maskw``: EDC time
        weights (None: no mask), ``inv``: what the EDC terms are divided by beyond the weights (1 when the weights are
        pre-normalised).  ``train``: gradients into the flat buffer, [all-reduce,] Adam (``opt_step=False`` stops in front
        of the all-reduce: the caller runs it and the update).  ``mask_draw``: callable that fills ``maskw`` on the
        device (run on the EDC stream, off the path to the output stage).  ``tail``: callable run on the side stream
        behind the last reader of the batch's receiver indices (GraphedTrainStep fetches the next step's there).
        ``pipe``: a :class:`StepPipe` -- this call is one step of a chain captured into one graph in which the side
        stream runs ahead (see the class).  Returns the loss dict of
        ``BandBankTrainer._step_losses`` (+ '_total').

        Scheduling rules (measured on the replayed graph, profiles/): a dependency that crosses streams costs
        8-12 us when the waiting stream is idle at the moment of the signal and nothing when the signal came earlier;
        and at a fork the graph keeps the FIRST captured successor on the producer's hardware queue.  So the critical
        chain stays on the main stream, its next kernel is captured before anything is forked off a node, and the
        side branches are arranged to finish early."""
        tr = self.tr
        bank, cfg, nb = tr.net, tr.config, tr.num_bands
        G, n = bank.num_groups, bank.num_delay_lines_per_group
        z, rows = data['z_values'], data['row_index']
        Btot = rows.numel()
        if Btot % nb:
            raise ValueError("the batch must hold the same number of receivers for every band")
        K = z.shape[-1]
        keep = self._keep
        main = torch.cuda.current_stream()
        side, side2 = tr._stream('_side'), tr._stream('_side2')
        on_side = (lambda: torch.cuda.stream(side)) if side is not None else _null
        on_side2 = (lambda: torch.cuda.stream(side2)) if side2 is not None else _null
        M = bank._blocks().detach()
        b, c = bank.input_gains.data.view(-1), bank.output_gains.data.view(-1)
        delays, ig = bank.delays, bank.inv_gamma
        gridK = FrequencyGrid.of(z)
        order = ops.irfft_slot_order(K, z.device) if (tr.use_slot_order and 'dataset' in data) else None
        Ku = (K + 1) // 2 if K % 2 == 1 else K
        # (receiver counts beyond what the sums over a band's receivers take fall through to the stored-signal chain)
        lin = (self.linear_transforms and pipe is None and ops.lin_supported(Btot // nb, G)
               and hasattr(data.get('dataset'), 'direct_time') and not self.fold_gains)
        if lin:
            zu, direct = (data['dataset'].slot_grid(*order) if order is not None else z[:Ku]), None
        elif order is not None:
            zu, direct = data['dataset'].slot_ordered(*order)
        else:
            zu, direct = z[:Ku], data['target_early_response'][:, :Ku]
        inv_world = shard_loss_scales(tr.world_size, 1, 1.0)['colorless']
        Hh, n_hidden, _, lo, hi = bank._mlp_cfg
        w = bank.output_scalars_w.detach()
        start, length = tr._decay_window(K)
        # bands whose longest decay times differ have their own EDC windows (reference trainer.py:56-59): per-item window
        # lengths for the EDC scans, ``maskw`` then holds one pre-normalised row per band
        item_len, _ = tr._item_windows(K, Btot // nb, z.device)
        if item_len is not None and (maskw is None or maskw.ndim != 2):
            raise ValueError("bands with different EDC windows: pass the (bands, window) weight rows of "
                             "BandBankTrainer._band_mask_rows as maskw")
        edr_t, edc_t = data['edr_target'], data['edc_target']
        T_edr, sum_abs, T_edc = edr_t[1], edr_t[2], edc_t[1]
        win = tr.stft_win
        pairs = order is not None and tr.use_pairs and win == 4096
        ev = {k: torch.cuda.Event() for k in ('start', 'g', 'h', 'mlp', 'mask', 'norm', 'x', 'edc', 'side', 'grg',
                                              'mlpb')}

        # ---- head.  main: records of the raw blocks -> energy pass -> finish (normalize, trainer.py:317-332);
        # side: rotations + records of the damped loop (taken at the gains BEFORE the rescale and scaled afterwards:
        # T is bilinear in b, c, T(b', c') = scale T(b, c)); side2: receiver gains (gain_filters.py:497-536), mask
        # (the head is a serial chain on the main stream: forking the rotations off it and joining again costs more
        # than the 10 us they take.  Captured BEFORE the fork below: the graph lays its hardware queues out along a
        # depth-first walk of the nodes in capture order, and the chain captured first keeps its queue through every
        # later join -- a chain that changes queue pays ~10 us per change)
        # blocks of 5..8 lines: the same step on the 17-polynomial records of csrc/blocktf8.hip (evaluation on the matrix
        # cores; unit-circle grids)
        big = n > 4
        if big and (gridK.logr is not None):
            raise NotImplementedError("blocks of more than four lines: the explicit step takes grids on the unit circle "
                                      "(set BandBankTrainer.use_fused = False to step this bank through the per-bin "
                                      "elimination kernels under autograd)")
        tail_ok = (self.fused_tail and train and opt_step and allreduce is None and pipe is None and side2 is not None
                   and self.adam_on_side)
        use_tail = tail_ok and not big
        # the normalisation scale joins the group signals behind the transform: energy pass on the side stream
        late = (self.scale_late and lin and order is not None and normalize_first and not big and side2 is not None)
        # blocks of 5..8 lines, integer delay lengths on the reference's own grid: normalize, the colorless pass and both
        # adjoints as real transforms of the coefficient sequences (csrc/polyfft.hip), normalize and the colorless pass on the
        # side stream; the forward group responses stay on the matrix-core pass (the dB stages of the decay losses need its
        # rounding: see that file).  The scale joins behind the transform, as ``late`` for the 4-line blocks
        tfp = None
        if (big and self.transform_polys and lin and order is not None and gridK.rfft_nfft and side2 is not None
                and normalize_first):
            T_seq = ops.tfp_plan(delays, n, gridK.rfft_nfft)
            sob = ops.tfp_slot_of_bin(K, z.device)
            if T_seq is not None and sob is not None:
                tfp = (gridK.rfft_nfft, sob, T_seq)
        late8 = tfp is not None
        use_tail8 = tail_ok and late8                 # (8-line blocks: csrc/blocktf8.hip k_tf8_tail)
        if not late and not late8:
            # (the side stream's head -- receiver gains, mask -- needs nothing of this step: forked off BEFORE the records launch,
            # so that the main chain's launches below stay first in capture order and keep its hardware queue)
            ev['start'].record()
        c_head = c
        if big:
            if use_tail8 and self.records_ok():
                # (the previous step's tail left the rotations of the current blocks and the snapshot of the output gains)
                Q, QQ, c_head = self._records()
            else:
                Q, QQ = ops.ortho_fwd(M, True, True)
                if late8:
                    # the side stream's normalize rescales the gains in place while the main stream's pass reads them: a
                    # snapshot
                    c_head = c.detach().clone()
                    keep.append(c_head)
            self._rec_valid = False           # (until this step's end says otherwise)
            coef, coef_sub = ops.tf8_coefs(QQ, ig, b, c, A1=M)
        else:
            # (the fused tail of the previous training step left them; anything else that touched M, b, c since made
            # the bookkeeping say so)
            # (a training step WITHOUT the fused tail always evaluates them: captured into a graph it would otherwise replay on
            # records nobody refreshes)
            if not (self.records_ok() and (use_tail or not train)):
                ops.tf_ortho_coefs(M, ig, b, c, out=self._records())
            Q, QQ, coef, coef_sub = self._records()
            self._rec_valid = False           # (until this step's end says otherwise)
        if late or late8:
            ev['start'].record()              # (the side stream's energy pass reads the raw blocks' records)
        spec_ok = (lin and self.spectral_edr and pairs and order is not None and win == 4096
                   and ops.spec_supported(Btot // nb, G) and hasattr(data['dataset'], 'direct_stft'))
        # the normalisation scale inside the receiver gains (scale_in_gains above)
        gfold = ((late or late8) and self.scale_in_gains and pipe is None and spec_ok and train
                 and self.gamma_dots_one_launch and self.gain_rows_in_mlp
                 and ops.mlp_bwd_takes_parts(bank._freq_pi.numel(), Hh, n_hidden, G, Btot // nb))
        if pipe is not None and (not train or allreduce is not None or not opt_step or side2 is None):
            raise ValueError("a pipelined step is a single-process training step with its optimiser update")
        scale = None
        Hg = Ts = None
        gridU = FrequencyGrid.of(zu)
        filt = tr._filter_on(Ku, order)
        if late:
            # the main chain's first launch is captured BEFORE anything is forked off: the graph lays its hardware queues out
            # along the capture order, and a chain whose head is captured behind a side stream's first launch starts every
            # replay on the second queue (measured: 29 us between the last kernel of a step and the group responses of the
            # next one, against ~6)
            Hg, Ts = ops.tf_compose_fwd(gridU.turns, gridU.logr, coef, delays, n, self._eye_rows(nb, G, z.device), None, None,
                                        filt, None, nb, save_T=True, want_H=True)
        Dinv8 = None
        if late8:
            # (as above: captured before the fork.  The grid in bin order -- T and 1 / Q where the adjoint's spectra are formed
            # --, the group responses through the band's filter scattered to the transform's slot order; unscaled: the
            # snapshot of the gains)
            Ts, _, Hg, Dinv8 = ops.tf8_tsave(gridK.turns[:Ku], coef, delays, n, c_head, None, nb, G, quad=False, filt=filt,
                                             want_H=True, hslot=tfp[1])
            keep.append(Dinv8)
        ework = None
        if normalize_first and not late and not late8:
            # (captured in front of the side stream's first launch: see above)
            if big:
                _, scale = ops.tf8_energy(gridK.turns, coef_sub, delays, n, b, c, dturn=gridK.dturn)
            else:
                _, scale = ops.tf_energy(gridK.turns, gridK.logr, coef_sub, delays, n, b, c, want_energy=False,
                                         dturn=gridK.dturn)
        if not late and not late8:
            ev['norm'].record()
        Xsub = None
        with on_side2():
            if pipe is None or pipe.first:
                torch.cuda.current_stream().wait_event(ev['start'])
            rgain0 = None
            if gfold and big:
                # 8-line blocks: the gain network FIRST (it needs nothing of this step), the transforms and the energy pass behind
                # it; the finish launch stores the gains times the scale -- rgain, what the step's launches take; rgain0, the
                # network's own output, is what its backward reads.  (4-line blocks: in front of the energy pass it measured
                # 0.355 against 0.352 ms, the same as no fold at all)
                rgain0, xhat, rstd = ops.mlp_gains_fwd(data['norm_listener_position'], bank._freq_pi, w, Hh, n_hidden,
                                                       G, lo, hi, rows, nb)
                keep.append(rgain0)
            if late:
                if gfold:
                    # (4-line blocks: the pass over the bins, the gain network, then the finish with the gains.  A bank with
                    # the reference's per-band networks -- three bands of 3 x 128 neurons -- runs the network FIRST: its
                    # launch is the longer of the two (33 us alone, 76 beside the energy pass and the transform passes when
                    # it started behind the pass: the EDR launch waited 45 us for the gains))
                    net_first = self.network_first if self.network_first is not None else bool(getattr(bank, 'mixed_networks', False))
                    if net_first:
                        rgain0, xhat, rstd = ops.mlp_gains_fwd(data['norm_listener_position'], bank._freq_pi, w, Hh,
                                                               n_hidden, G, lo, hi, rows, nb)
                    ework = ops.tf_energy(gridK.turns, gridK.logr, coef_sub, delays, n, b, c, phase=1, dturn=gridK.dturn)
                    if not net_first:
                        rgain0, xhat, rstd = ops.mlp_gains_fwd(data['norm_listener_position'], bank._freq_pi, w, Hh,
                                                               n_hidden, G, lo, hi, rows, nb)
                    keep.append(rgain0)
                    _, scale, rgain = ops.tf_energy(gridK.turns, gridK.logr, coef_sub, delays, n, b, c, want_energy=False,
                                                    work=ework, phase=2, dturn=gridK.dturn, gains=rgain0, G=G)
                else:
                    _, scale = ops.tf_energy(gridK.turns, gridK.logr, coef_sub, delays, n, b, c, want_energy=False,
                                             dturn=gridK.dturn)
                ev['norm'].record()
            elif late8:
                nblk8 = nb * G
                Xsub = ops.tfp_forward(coef_sub, delays, c, n, tfp[0], tfp[2])
                if gfold:
                    _, scale, rgain = ops.tfp_energy(Xsub[:nblk8], Xsub[nblk8:], n, b, c, gains=rgain0, G=G)
                else:
                    _, scale = ops.tfp_energy(Xsub[:nblk8], Xsub[nblk8:], n, b, c)
                ev['norm'].record()
                keep.append(Xsub)
            if gfold:
                ev['mlp'].record()
            elif pipe is None:
                rgain, xhat, rstd = ops.mlp_gains_fwd(data['norm_listener_position'], bank._freq_pi, w, Hh, n_hidden,
                                                      G, lo, hi, rows, nb)
                ev['mlp'].record()
            else:
                # the previous step of the chain (or the prologue) has left this step's receiver gains in the pipe
                rgain, xhat, rstd = pipe.cur
            # the EDC time mask is drawn on the stream that runs the EDC scans.  (Drawn on `side`, with the reported
            # total on `side` waiting for the sums of `side2`, the two forked streams depend on each other in both
            # directions -- hipStreamEndCapture of ROCm 7.2 segfaults on that topology.)
            if mask_draw is not None:
                mask_draw()
            ev['mask'].record()

        def wait_gains():
            if pipe is None:
                main.wait_event(ev['mlp'])
            elif pipe.ready is not None:
                main.wait_event(pipe.ready)

        fold = (self.fold_output_stage and pairs and K == 65537 and (Btot // nb) % 2 == 0 and G <= 4
                and not lin)
        # the receiver gains come from the side stream: with the output stage folded into the transform the transfer-function
        # launch below does not read them, and the wait goes behind it -- by then the event was signalled long ago (a wait on
        # a signalled event is free; in front of the launch the idle main stream paid a cross-queue wake-up of ~12 us)
        if not fold and not lin:
            wait_gains()
        x_fn = None
        tau = eye = None
        spec = False
        tau_pairs = order is not None
        if lin:
            # group responses through the band's filter -> G time signals per band; the receivers' signals are formed from
            # them and the dataset's transformed direct paths in one streaming pass (csrc/linear.hip)
            eye = self._eye_rows(nb, G, z.device)
            xd = data['dataset'].direct_time(tr.subband_filter_freq_resp, K)
            if late8:
                pass                          # (launched above)
            elif big:
                # (the group responses through the band's filter written by the same launch: no tensor operation between the
                # transfer functions and the transform)
                Ts, _, Hg, Dinv8 = ops.tf8_tsave(gridU.turns, coef, delays, n, c, scale, nb, G, quad=False, filt=filt,
                                                 want_H=True)
                keep.append(Dinv8)
                ev_ts = torch.cuda.Event()
                ev_ts.record()
            elif not late:
                Hg, Ts = ops.tf_compose_fwd(gridU.turns, gridU.logr, coef, delays, n, eye, scale, None, filt, None, nb,
                                            save_T=True, want_H=True)
            if gfold:
                tau = ops.irfft_odd_fwd(Hg, K, slots=True, pairs=True)     # (signals of the unscaled functions: no wait)
            elif late or late8:
                # (the wait sits in front of the LAST pass, the only reader of the scale)
                tau = ops.irfft_odd_fwd(Hg, K, slots=True, pairs=True, oscale=scale,
                                        before_last=lambda: main.wait_event(ev['norm']))
            else:
                tau = ops.irfft_odd_fwd(Hg, K, slots=order is not None, pairs=tau_pairs)
            H = Hg
            spec = spec_ok

            def x_fn():
                wait_gains()
                return ops.lin_combine_fwd(xd, rows, tau, rgain, nb, K, tau_pairs, pairs)
            keep.extend((tau, xd))
        elif big:
            Ts, Tq8 = ops.tf8_tsave(gridU.turns, coef, delays, n, c, scale, nb, G, quad=True)
            # (the colorless pass of the 8-line blocks is VALU-bound like this launch: it starts behind it and runs
            # beside the memory-bound transform instead -- measured: tsave 105 -> 55 us in the step)
            ev_ts = torch.cuda.Event()
            ev_ts.record()
            if fold:
                H = Tq8
            else:
                # (sizes the folded transform does not take: the output stage as tensor operations -- small grids only)
                Bb = Btot // nb
                Hs = torch.einsum('qbg,qgk->qbk', rgain.reshape(nb, Bb, G).to(torch.complex64), Ts.reshape(nb, G, -1))
                Hs = Hs + direct[rows].reshape(nb, Bb, -1)[..., :Ts.shape[1]]
                if filt is not None:
                    Hs = Hs * filt.reshape(nb, 1, -1)
                H = Hs.reshape(Btot, -1).contiguous()
        else:
            H, Ts = ops.tf_compose_fwd(gridU.turns, gridU.logr, coef, delays, n, rgain, scale, direct, filt, rows, nb,
                                       save_T=True, want_H=not fold)
        if fold:
            Tq, H = H, None

            def x_fn():
                wait_gains()
                x2, h0 = ops.irfft_odd_pairs_compose_fwd(direct, rows, Tq, rgain, filt, K, nb)
                keep.extend((h0, Tq))
                return x2
        keep.extend((coef_sub, ework, Q, QQ, coef, rgain, xhat, rstd, scale, H, Ts))

        # ---- colorless pass (spectral loss + dL/drecords of the sub-FDNs, sparsity): on the EDC stream.  Blocks of <= 4
        # lines: in front of the scans, beside the output stage and the transform (behind the scans it ran beside the STFT
        # adjoint: same step time either way, measured).  Blocks of 5..8 lines: BEHIND the scans -- the pass takes ~200 us
        # there, and in front of them the EDC gradient reached the odd-frame launch of the STFT adjoint late

        def colorless_pass():
            with on_side2():
                if not late and not late8:         # (recorded on this very stream: no wait -- a captured wait of a stream
                    torch.cuda.current_stream().wait_event(ev['norm'])      # for its own event is asking for trouble)
                if late8:
                    grec_sub_, loss_g_ = ops.tfp_colorless(Xsub[:nb * G], Xsub[nb * G:], tfp[0], n, delays, scale,
                                                           cfg.use_asym_spectral_loss, cfg.spectral_loss_weight * inv_world,
                                                           T=tfp[2])
                elif big:
                    torch.cuda.current_stream().wait_event(ev_ts)
                    grec_sub_, loss_g_ = ops.tf8_colorless(gridK.turns, coef_sub, delays, n, c, scale,
                                                           cfg.use_asym_spectral_loss,
                                                           cfg.spectral_loss_weight * inv_world, dturn=gridK.dturn)
                else:
                    grec_sub_, loss_g_ = ops.tf_colorless(gridK.turns, gridK.logr, coef_sub, delays, n, scale,
                                                          cfg.use_asym_spectral_loss,
                                                          cfg.spectral_loss_weight * inv_world, dturn=gridK.dturn)
                out3_, gQ_ = ops.colorless_terms(loss_g_, Q, cfg.spectral_loss_weight, cfg.sparsity_loss_weight,
                                                 inv_world, want_grad=train, nbands=nb)
                ev['side'].record()
            keep.extend((grec_sub_, loss_g_, out3_, gQ_))
            return grec_sub_, out3_, gQ_

        # (spectral-EDR step: the same holds for the small blocks -- in front of the scans the VALU-bound pass ran beside the
        # latency-bound transforms of the G group signals, which sit on the critical chain: 18.6 against 9.6 us for the
        # row pass; behind them it runs beside the memory-bound EDR kernels)
        late_colorless = (big or (spec and self.colorless_late_small)) and self.colorless_behind_scans
        if not late_colorless:
            grec_sub, out3, gQ = colorless_pass()

        # ---- decay losses: irfft -> STFT -> EDR -> STFT adjoint -> irfft adjoint, EDC scans beside them
        ev['h'].record()
        spec_bw = None
        if spec:
            li_edr, li_edc, spec_bw = self._decay_middle_spec(data, K, rows, rgain, tau, xd, maskw, inv, train, T_edr,
                                                              sum_abs, T_edc, start, length, item_len, nb, G, ev, main,
                                                              side2, wait_gains)
            gH = None
        else:
            li_edr, li_edc, gH = self._decay_middle(H, K, rows, maskw, inv, train, order, pairs, T_edr, sum_abs, T_edc,
                                                    start, length, ev, main, side2, x_fn=x_fn, Btot=Btot,
                                                    gains=(Tq, filt, nb, G) if (fold and train and self.fold_gains) else None,
                                                    item_len=item_len, Bper=Btot // nb, lin=lin)
        if late_colorless:
            grec_sub, out3, gQ = colorless_pass()
        def report():
            """the reported sums and total (off the gradient path)"""
            s_ = ops.weighted_sums(li_edr, cfg.edr_loss_weight, li_edc, cfg.edc_loss_weight, sum_abs, rows, nb)
            return s_, ((s_[:, 0] + out3[:, 0]) if nb > 1 else (s_[0] + out3[0]))

        if train:
            if (use_tail or use_tail8) and not torch.cuda.is_current_stream_capturing():
                tr.optimizer.sync_lr()            # (both Adam launches of the split update read the device table)
            # ---- backward of the output stage.  Its two passes are independent and both stream dL/dH: the records pass
            # stays on the main stream, the gains pass (-> gain network backward) runs beside it on side2 -- together
            # 50 us instead of 29 + 45 one after the other.  The reported sums go in front of the gains pass (their
            # inputs are long complete; behind it they would sit on the path to Adam).
            ev['grg'].record()                    # (dL/dH complete)
            if lin:
                # time-domain adjoint: gamma_g = sum_b rgain[b][g] dL/dx[b] (G signals per band) -> their adjoint transform
                # = dL/d(T_g filt) -> the records pass at "B = G receivers with identity gains"
                sot = ops.irfft_slot_of_time(K, z.device) if tau_pairs else None
                if spec:
                    # dL/dtau = [EDC part: sum_b rgain dL/dx_b of the scans] + [EDR part: adjoint STFT of the band's G
                    # gradient spectra, already summed over the receivers]
                    gsig, gsig_b, gam_edr, parts = spec_bw[0], None, spec_bw[1], spec_bw[2]
                    main.wait_event(ev['edc'])
                    if self.gamma_dots_one_launch:
                        # one sweep over the EDC gradient signals (their window only): the G sums per band AND the EDC part
                        # of dL/drgain
                        _, band_len = tr._item_windows(K, Btot // nb, z.device)
                        ge_a, ge_b = gam_edr if isinstance(gam_edr, tuple) else (gam_edr, None)
                        if isinstance(gsig, tuple):         # (rows, their sums per group: bankstep.gamma_split)
                            gam = ops.lin_merge_slots(gsig[1], ge_a, ge_b, sot)
                        elif self._edc_one(length, G):
                            gam = ops.lin_gamma_win(gsig, rgain, nb, K, start, length, base=ge_a, base_b=ge_b,
                                                    slot_of_time=sot, band_win_len=band_len)
                        else:
                            gam = ops.lin_gamma_dots(gsig, rgain, nb, K, tau, parts, start, length, base=ge_a,
                                                     slot_of_time=sot, band_win_len=band_len, base_b=ge_b)
                    else:
                        gam = ops.lin_gamma(gsig, rgain, nb, K, True, True, slot_of_time=sot, base=gam_edr)
                    ev_gam = torch.cuda.Event()
                    ev_gam.record()
                else:
                    gsig, gsig_b = gH
                    gam = ops.lin_gamma(gsig, rgain, nb, K, pairs, tau_pairs, gxb=gsig_b, slot_of_time=sot)
                if tau_pairs:
                    gHg = ops.irfft_odd_pairs_bwd(gam, K, nb * G, tslots=sot is not None)
                else:
                    gHg = ops.irfft_odd_bwd(gam, K, Ku, None, slots=False)
                keep.extend((gam, gHg, eye))
                gH_rec, rg_rec = gHg, eye
            else:
                gH_rec, rg_rec = gH, rgain
            if late8:
                grec = ops.tfp_compose_bwd(tfp[0], nb, G, n, delays, Ku, tfp[1], gH_rec, filt, Ts, Dinv8, tscale=scale,
                                           gain_fold=gfold, T=tfp[2])
            elif big:
                # (the linear step's adjoint runs on the grid of the forward pass: its saved T' and 1 / Q come back)
                grec = ops.tf8_compose_bwd(gridU.turns, coef, delays, n, c, scale, rg_rec, gH_rec, filt, nb,
                                           saved=(Ts, Dinv8) if lin else None)
            else:
                grec = ops.tf_compose_bwd(gridU.turns, gridU.logr, coef, delays, n, rg_rec, gH_rec, Ts, filt, nb, partial=True,
                                          tscale=scale if late else None, gain_fold=gfold and late)
            with on_side2():
                if spec and self.gamma_dots_one_launch:
                    torch.cuda.current_stream().wait_event(ev_gam)           # (behind the EDR launch as well)
                else:
                    torch.cuda.current_stream().wait_event(ev['g'])
                sums, total = report()
                if allreduce is not None:
                    # data-parallel: this rank's loss terms ride the gradient bucket -- [EDR | EDC | colorless share]
                    # per band behind the gradients -- and ONE all-reduce sums both over the ranks.
                    # Written here, off the path: the main stream's wait for this branch covers it
                    slots = tr.optimizer.extra.view(3, nb)
                    torch.stack((sums.reshape(nb, 3)[:, 1], sums.reshape(nb, 3)[:, 2], out3.reshape(nb, 3)[:, 0]),
                                out=slots)
                if not lin:
                    torch.cuda.current_stream().wait_event(ev['grg'])
                if lin:
                    # dL/drgain[b][g] = <dL/dx[b], tau_g>: partial rows over the time samples
                    if spec:          # (the EDR part of the rows was left by the EDR kernel; the EDC part joins it)
                        # behind the gamma pass: both stream the same 59 MB, and the gamma pass heads the critical chain
                        # (side by side: 40 us for it instead of 15)
                        torch.cuda.current_stream().wait_event(ev_gam)
                        ggp = parts if self.gamma_dots_one_launch else \
                            ops.lin_gain_dots(gsig, tau, nb, Btot, G, K, True, True, out=parts)
                    else:
                        ggp = ops.lin_gain_dots(gsig, tau, nb, Btot, G, K, pairs, tau_pairs, gxb=gsig_b)
                    grg = None
                    if not (self.gain_rows_in_mlp
                            and ops.mlp_bwd_takes_parts(bank._freq_pi.numel(), Hh, n_hidden, G, Btot // nb)):
                        grg, ggp = ops.tf_rows_sum(ggp).view(Btot, G), None
                elif fold and self.fold_gains:
                    grg = ops.tf_rows_sum(self._gpart)
                    ggp = None
                elif self.gain_rows_in_mlp and ops.mlp_bwd_takes_parts(bank._freq_pi.numel(), Hh, n_hidden, G, Btot // nb):
                    # the gains pass leaves its per-chunk partial rows; every receiver's wave of the gain network's
                    # backward sums its own (same order as the row-sum launch that sat between the two, on the path to
                    # the update)
                    grg, ggp = None, ops.tf_gain_grad(Ts, gH, G, filt, nb, partial=True)
                else:
                    grg, ggp = ops.tf_gain_grad(Ts, gH, G, filt, nb), None
                ops.mlp_gains_bwd(data['norm_listener_position'], bank._freq_pi, w, Hh, n_hidden, G, lo, hi,
                                  rgain0 if gfold else rgain, xhat, rstd, grg, rows, nb, out=self.g_w, ggains_parts=ggp,
                                  colscale=scale if gfold else None)
                keep.append(ggp)
                if pipe is not None:
                    # the gain network's range of the flat buffers is stepped HERE, behind its gradient, on this
                    # stream; then the next step's receivers and its receiver gains -- the main stream never waits
                    # for this branch again until the next output stage needs them
                    tr.optimizer.step_range(*self.w_range, second=True)
                    if tail is not None:
                        tail()
                    ops.mlp_gains_fwd(data['norm_listener_position'], bank._freq_pi, w, Hh, n_hidden, G, lo, hi, rows,
                                      nb, out=pipe.nxt)
                    pipe.ready_next = torch.cuda.Event()
                    pipe.ready_next.record()
                else:
                    if use_tail or use_tail8:
                        # the gain network's range of the flat buffers, straight behind its gradient on this stream (its
                        # own step counter: no ordering with the main stream's fused tail)
                        tr.optimizer.step_range(*self.w_range, second=True)
                    ev['mlpb'].record()
                    if tail is not None:
                        tail()            # (every reader of ``rows`` is ordered before this point: main's are in
                                          # front of ev['grg'], this stream's are its own earlier launches)
            main.wait_event(ev['side'])          # (signalled long ago; dropping it measured no gain: 0.667 vs 0.663 ms)
            # records (partial rows of the records pass + the colorless pass's) -> dL/dM, dL/db, dL/dc: one launch
            if use_tail8:
                # ... -> Adam on the blocks' own M, b, c -> the NEXT step's Q, Q Q and gain snapshot, in the same launch
                ops.tf8_tail(QQ, ig, grec, grec_sub, b, c, M, gQ, Q, self.g_b, self.g_c, self.g_M, tr.optimizer, *self._off,
                             *self._records())
            elif big:
                ops.tf8_param_grads(QQ, ig, grec, b, c, M, A1=M, part1=grec_sub, gQ=gQ, Q=Q, gb=self.g_b, gc=self.g_c,
                                    gM=self.g_M)
            elif use_tail:
                # ... -> Adam on the blocks' own M, b, c -> the NEXT step's Q, QQ and record sets, in the same launch
                ops.tf_tail(QQ, ig, grec, grec_sub, b, c, M, gQ, Q, self.g_b, self.g_c, self.g_M.view(-1), tr.optimizer,
                            *self._off, *self._records())
            else:
                ops.tf_param_grads(QQ, ig, grec, b, c, M, A1=M, grec1=grec_sub, gQ=gQ, Q=Q, gb=self.g_b, gc=self.g_c,
                                   gM=self.g_M)
            keep.extend((grg, grec))
            if pipe is not None:
                # everything but the gain network, straight behind its gradients
                tr.optimizer.step_range(0, self.w_range[0], second=False)
                torch.autograd.graph.increment_version(tr.optimizer._params)
                sums_total = (sums, total)
                return self._finish_pipe(pipe, sums_total, out3, nb, main, side, side2)
            tr.optimizer._packed = True               # the flat gradient buffer is complete
            if use_tail or use_tail8:
                # both ranges of the flat buffers are stepped (main: fused tail; side2: the gain network's own launch)
                tr.optimizer._packed = False
                torch.autograd.graph.increment_version(tr.optimizer._params)
                self._rec_valid, self._rec_versions = True, self._versions()
            elif opt_step and allreduce is None and side2 is not None and self.adam_on_side:
                # single process: the update runs on the branch that finishes LAST (the gain network's backward), behind
                # an event of the main stream that was signalled earlier -- a wait on a long-signalled event is free,
                # while the main stream, idle when the side branch signals, would pay the 8-12 us of a cross-queue wake-up
                ev_pg = torch.cuda.Event()
                ev_pg.record()
                with on_side2():
                    torch.cuda.current_stream().wait_event(ev_pg)
                    self.finish(None)
            else:
                main.wait_event(ev['mlpb'])
                if opt_step:
                    red = self.finish(allreduce)
                    if red is not None:
                        sums, total = red
        else:
            with on_side2():
                torch.cuda.current_stream().wait_event(ev['g'])
                sums, total = report()
        keep.extend((sums, total))
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
        if not big and not train and not normalize_first:
            self._rec_valid, self._rec_versions = True, self._versions()      # (nothing touched M, b, c)
        return losses


    def _finish_pipe(self, pipe, sums_total, out3, nb, main, side, side2):
        """End of one step of a pipelined chain: no join with the side stream (only the chain's last step joins, which
        stream capture needs); this step's tensors stay referenced until the END of the next step -- by then every
        stream has passed a point ordered behind their last readers (the allocator hands a freed block only to the
        stream it was allocated on, but main-stream blocks are read on the side stream too)."""
        sums, total = sums_total
        keep = self._keep
        keep.extend((sums, total))
        if nb > 1:
            losses = {'edc_loss': sums[:, 2], 'edr_loss': sums[:, 1], 'spectral_loss': out3[:, 1],
                      'sparsity_loss': out3[:, 2], '_total': total}
        else:
            losses = {'edc_loss': sums[2], 'edr_loss': sums[1], 'spectral_loss': out3[1], 'sparsity_loss': out3[2],
                      '_total': total}
        if pipe.last:
            for s_ in (side, side2):
                if s_ is not None:
                    main.wait_stream(s_)
            self._keep_prev = []
            keep.clear()
        else:
            self._keep_prev = list(keep)
            keep.clear()
        pipe.advance()
        return losses

    # -- the part of a data-parallel step behind the gradients ------------------------------------------------------
    def finish(self, allreduce):
        """[all-reduce of the bucket,] Adam.  Returns (sums, total) rebuilt from the reduced loss slots -- the whole
        job's EDR / EDC terms and total per band -- or None for a single process."""
        tr = self.tr
        nb = tr.num_bands
        red = None
        if allreduce is not None:
            allreduce()
        tr.optimizer.step()                  # (straight behind the collective: the reported sums below are not on the path)
        if allreduce is not None:
            slots = tr.optimizer.extra.view(3, nb)
            total = slots.sum(dim=0)
            sums = torch.stack((total, slots[0], slots[1]), dim=1)          # [total, w_edr edr, w_edc edc] per band
            red = (sums, total) if nb > 1 else (sums[0], total[0])
        return red


class StepPipe:
    """State of a chain of explicit steps captured into ONE graph with the side stream running ahead of the main one.

    In a single step the main stream waits for the side stream twice where nothing but latency is gained or lost: at
    the head (the receiver gains of the batch, ~12 us behind the side stream's fork) and at the tail (Adam behind the
    gain network's backward, ~15 us), and every step ends with a join and starts with a fork (~11 us).  In a chain the
    side stream steps the gain network's parameters itself, fetches the next step's receivers and evaluates their gains
    right behind the gain network's backward of the current step; the main stream steps everything else straight
    behind its own last gradient kernel and goes on with the next step.  The numbers are those of the single steps,
    bit for bit (tests/test_gpu_bank.py).

    ``bufs``: two sets of (gains (B, G), xhat (B, nl, H), rstd (B, nl)) owned by the caller: step i of the chain reads
    set i % 2 and fills set (i + 1) % 2, so a chain of EVEN length leaves the next chain's first set filled."""

    def __init__(self, bufs, steps: int):
        if steps % 2:
            raise ValueError("a pipelined chain has an even number of steps")
        self.bufs, self.steps, self.i = bufs, steps, 0
        self.ready = None            # event behind the gains of the CURRENT step (None: produced before this chain)
        self.ready_next = None

    @property
    def cur(self):
        return self.bufs[self.i % 2]

    @property
    def nxt(self):
        return self.bufs[(self.i + 1) % 2]

    @property
    def first(self):
        return self.i == 0

    @property
    def last(self):
        return self.i == self.steps - 1

    def advance(self):
        self.i += 1
        self.ready, self.ready_next = self.ready_next, None


class _null:
    """no-op context (a branch that has no stream of its own runs inline on the main stream)"""

    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
