"""Band bank: the octave-band GFDNs of one room stepped in lockstep, one launch per stage for ALL bands.

The reference trains one independent DiffGFDNVarReceiverPos per octave band, one after another
(src/run_subband_training_treble.py:175-204: ``for k in range(len(freqs_list))`` builds the band's
config, dataset, model and ``VarReceiverPosTrainer`` and calls ``train``).  The bands share neither
parameters nor data, only their SHAPE: same number of groups and delay lines per group, same MLP
sizes, same frequency grid, same batch size.  On an MI355X a single band at batch 32 fills a
fraction of the 256 CUs (the step is a dependency chain of small launches), so the
bank evaluates the same chain ONCE with every stage widened over the bands:

  * the per-bin solve takes the bands' diagonal blocks side by side (nblk = bands x G blocks of one
    ``gfdn_solve_*`` launch; delays / absorption gains / input gains are per-lane arrays anyway);
  * the output stage, the gain network, the colorless bookkeeping and the loss totals run their
    band-stacked entry points (``gfdn_*_banded``, include/diffgfdn_hip.h): items are band-major
    (item = band x B + b) and every band reads its own parameter row;
  * irfft / STFT / EDR / EDC are per-item kernels already: they see bands x B items;
  * all bands' parameters live in ONE flat Adam buffer (one update launch, one all-reduce).

Results per band are those of the band's own trainer (tests/test_gpu_bank.py checks losses, gradients
and the post-Adam state of every band against single-band steps and the CPU oracle).  Each band's
``nn.Module`` stays alive: its parameters are views into the bank's stacked tensors, so
``nets[i].state_dict()`` is the reference-shaped checkpoint of band i at any time.

Supported layout (the reference's sub-band recipe, run_subband_training_treble.py:105-154): SCALAR
coupling with zero inter-group coupling, fixed common decay times, MLP output gains.  Other layouts
train band by band with ``subband.train_bands``.
"""
import contextlib
import os
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from . import hip_ops as ops
from .config import CouplingMatrixType, TrainerConfig
from .functional import (ColorlessTerms, FrequencyGrid, MlpGains, OrthoParam, OutputStage, ResolventSolve,
                         SubFdnColorless)
from .bankstep import FusedBankStep
from .losses import decay_losses, edc_loss, shard_loss_scales
from .optim import FlatAdam


class BandBank(nn.Module):
    """``nets``: one DiffGFDNVarReceiverPos per band, already on the GPU, identical in shape.

    Stacked leaves (what autograd and Adam see):
        input_gains, output_gains (bands, N);  feedback_loop_M (bands, G, n, n);
        output_scalars_w (bands, P) -- every band's MLP parameters packed in named_parameters() order (bands whose networks
        differ in size: ONE vector, band after band).
    The names keep the substrings the reference selects learning-rate groups by
    (trainer.py:157-216: 'input_gains', 'output_gains', 'output_scalars' -> io_lr, the rest -> lr)."""

    def __init__(self, nets: Sequence[nn.Module]):
        super().__init__()
        nets = list(nets)
        if not nets:
            raise ValueError("BandBank needs at least one band")
        n0 = nets[0]
        G, n = n0.num_groups, n0.num_delay_lines_per_group
        for net in nets:
            fl = net.feedback_loop
            if (net.num_groups, net.num_delay_lines_per_group) != (G, n):
                raise ValueError("BandBank: bands must have the same groups x delay lines per group")
            if not fl.uncoupled:
                raise NotImplementedError("BandBank: zero inter-group coupling only (train other layouts "
                                          "band by band with subband.train_bands)")
            if fl.use_absorption_filters:
                raise NotImplementedError("BandBank: scalar absorption gains (the sub-band recipe)")
            if fl.learn_decay_times or not isinstance(fl.M, nn.Parameter):
                raise NotImplementedError("BandBank: fixed decay times and a learnable M per band")
            if not hasattr(net, 'output_scalars') or not hasattr(net.output_scalars, 'mlp'):
                raise NotImplementedError("BandBank: DiffGFDNVarReceiverPos with MLP output gains")
            if not net.input_gains.is_cuda:
                raise RuntimeError("BandBank needs the band models on the GPU (no CPU fallback)")
            if net.sample_rate != n0.sample_rate:
                raise ValueError("BandBank: one sample rate")
        object.__setattr__(self, 'nets', nets)            # not sub-modules: no duplicate parameters
        self.num_bands = len(nets)
        self.num_groups, self.num_delay_lines_per_group = G, n
        self.num_delay_lines = G * n
        self.sample_rate = n0.sample_rate
        self.use_colorless_loss = all(net.use_colorless_loss for net in nets)
        N = G * n
        # every band's gain network: (neurons per layer, hidden layers).  The reference's sub-band driver sizes them per band
        # (run_subband_training_treble.py:61-73: 1 x 8, 1 x 16, 5 x 16, 3 x 128); features and output limits are shared
        cfgs = []
        for net in nets:
            gm = net.output_scalars
            lin = [m for m in gm.mlp.model if isinstance(m, nn.Linear)]
            if any(l.out_features != lin[0].out_features for l in lin[:-1]):
                raise NotImplementedError("BandBank: gain networks with one width for all hidden layers")
            cfgs.append((lin[0].out_features, len(lin) - 2, gm.encoder.num_fourier_features,
                         float(gm.scaled_sigmoid.lower_limit), float(gm.scaled_sigmoid.upper_limit)))
        if any(c[2:] != cfgs[0][2:] for c in cfgs):
            raise ValueError("BandBank: the bands' gain networks must share the Fourier features and the output limits")
        self.mixed_networks = any(c[:2] != cfgs[0][:2] for c in cfgs)
        # (H, hidden layers, F, lo, hi): ints for a bank of equal networks, one tuple entry per band otherwise -- the
        # kernels' wrappers (hip_ops.mlp_gains_*) take either
        self._mlp_cfg = ((tuple(c[0] for c in cfgs), tuple(c[1] for c in cfgs)) + cfgs[0][2:]) if self.mixed_networks \
            else cfgs[0]
        self._mlp_params = [[p for m in net.output_scalars.mlp.model for p in m.parameters()] for net in nets]
        with torch.no_grad():
            self.input_gains = nn.Parameter(torch.stack([net.input_gains.detach().reshape(N) for net in nets]))
            self.output_gains = nn.Parameter(torch.stack([net.output_gains.detach().reshape(N) for net in nets]))
            self.feedback_loop_M = nn.Parameter(torch.stack([net.feedback_loop.M.detach() for net in nets]))
            packed = [torch.cat([p.detach().reshape(-1) for p in ps]) for ps in self._mlp_params]
            # equal networks: (bands, P); mixed: the bands' packed sets one after the other in ONE vector
            self.output_scalars_w = nn.Parameter(torch.cat(packed) if self.mixed_networks else torch.stack(packed))
            self._w_off = [0]
            for t in packed:
                self._w_off.append(self._w_off[-1] + t.numel())
            self.register_buffer('delays', torch.cat([net.delay_buffer.to(torch.float32) for net in nets]),
                                 persistent=False)
            self.register_buffer('inv_gamma', torch.cat(
                [(1.0 / net.feedback_loop.delay_line_gains.to(torch.float32)) for net in nets]).contiguous(),
                persistent=False)
            dev = self.input_gains.device
            self.register_buffer('_ones', torch.ones(self.num_bands * N, dtype=torch.float32, device=dev),
                                 persistent=False)
            self.register_buffer('_eye', torch.eye(G, dtype=torch.float32, device=dev).repeat(self.num_bands, 1),
                                 persistent=False)
            f = torch.exp(torch.linspace(np.log(1.0), np.log(32.0), self._mlp_cfg[2], device=dev))
            self.register_buffer('_freq_pi', (f * np.pi).to(torch.float32).contiguous(), persistent=False)
        self.relink()

    def relink(self):
        """Point every band model's parameters at the bank's (current) storage.  Call again after
        anything that re-homes the leaves (FlatAdam moves them into its flat buffer)."""
        N = self.num_delay_lines
        for i, net in enumerate(self.nets):
            net.input_gains.data = self.input_gains.data[i].view(N, 1)
            net.output_gains.data = self.output_gains.data[i].view(N, 1)
            net.feedback_loop.M.data = self.feedback_loop_M.data[i]
            off = self._w_off[i]
            wflat = self.output_scalars_w.data.view(-1)
            for p in self._mlp_params[i]:
                k = p.numel()
                p.data = wflat[off:off + k].view(p.shape)
                off += k

    def band_state_dict(self, band: int) -> Dict[str, torch.Tensor]:
        """Reference-shaped state dict of one band (DiffGFDNVarReceiverPos.state_dict keys)."""
        return self.nets[band].state_dict()

    # -- stages, every one a single launch over all bands -----------------------------------------
    def _blocks(self) -> torch.Tensor:
        G, n = self.num_groups, self.num_delay_lines_per_group
        return self.feedback_loop_M.view(self.num_bands * G, n, n)

    @torch.no_grad()
    def normalize(self, z: torch.Tensor):
        """Trainer.normalize (trainer.py:317-332) for every band: b_n, c_n /= E_g^(1/4)."""
        grid = FrequencyGrid.of(z)
        ops.subfdn_normalize(grid.turns, grid.logr, self._blocks().detach(), self.delays,
                             self.input_gains.data.view(-1), self.output_gains.data.view(-1))
        # (the kernel rescales b, c through raw pointers: say so to whoever keeps derived state of them -- the explicit step's
        # records that outlive a step compare tensor versions, bankstep.FusedBankStep.records_ok)
        torch.autograd.graph.increment_version((self.input_gains, self.output_gains))

    def rotations(self):
        """(Q, QQ) (bands*G, n, n): Q_g = expm(skew(M_g)), QQ_g = Q_g Q_g (feedback_loop.py:393-404)."""
        return OrthoParam.apply(self._blocks())

    def sub_fdn_group_sums(self, z: torch.Tensor) -> torch.Tensor:
        """S (bands*G, K): un-damped sub-FDN responses Hout[k, g] of every band (model.py:209-252)."""
        grid = FrequencyGrid.of(z)
        Ysub = ResolventSolve.apply(self._blocks(), self._ones, self.input_gains.view(-1), grid, self.delays,
                                    False)
        return OutputStage.apply(Ysub, self.output_gains.view(-1), self._eye,
                                 self.num_delay_lines_per_group, None, None, None, self.num_bands)

    def delay_line_responses(self, z: torch.Tensor, QQ: torch.Tensor) -> torch.Tensor:
        """Y (K, bands*N): y = (D Gamma^-1 - blockdiag(Q_g Q_g))^-1 b for every band."""
        return ResolventSolve.apply(QQ, self.inv_gamma, self.input_gains.view(-1), FrequencyGrid.of(z),
                                    self.delays, False)

    def group_gains(self, positions: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
        """(bands*B, G) receiver gains; item i encodes positions[rows[i]] with band i // B's network."""
        H, n_hidden, _, lo, hi = self._mlp_cfg
        return MlpGains.apply(positions, rows, self._freq_pi, H, n_hidden, self.num_groups, lo, hi,
                              self.output_scalars_w)


class BandStackedDataset:
    """The bands' MultiRIRDataset stores stacked band-major: row = band * R + receiver.

    Same ``collate(rows, lean="rows")`` contract as MultiRIRDataset (dataloader.py here,
    custom_collate dataloader.py:674-704 in the reference): nothing is gathered, the batch is an
    int64 vector of global rows into the stores."""

    def __init__(self, datasets: Sequence, free_sources: bool = False):
        ds0 = datasets[0]
        self.device = ds0.device
        self.num_bands = len(datasets)
        self.R = len(ds0)
        for d in datasets:
            if len(d) != self.R or d.z_values.shape != ds0.z_values.shape:
                raise ValueError("BandStackedDataset: every band needs the same receivers and frequency grid")
        self.z_values = ds0.z_values
        self.datasets = list(datasets)
        self.norm_listener_position = torch.cat([d.norm_listener_position for d in datasets])
        self.listener_positions = torch.cat([d.listener_positions for d in datasets])
        self.early_rir_mag_response = torch.cat([d.early_rir_mag_response for d in datasets])
        et = [getattr(d, 'early_rir_time', None) for d in datasets]
        self.early_rir_time = torch.cat(et) if all(t is not None and t.shape[1] == et[0].shape[1] for t in et) else None
        self.rir_mag_response = ds0.rir_mag_response          # shape carrier (K); targets come from the stores
        self.source_position = ds0.source_position
        self.edr_store = self.edc_store = None
        if free_sources:
            for d in datasets:
                d.early_rir_mag_response = None

    def __len__(self):
        return self.R

    def precompute_decay_targets(self, win: int, edc_start: int, edc_len, chunk: int = 64):
        """``edc_len``: one window length for all bands, or one per band (bands whose longest decay times differ,
        reference trainer.py:56-59): the stacked EDC store then has rows of the LONGEST window, band q's rows filled up to
        its own length (the banded EDC kernels read them with that pitch), and its key is (start, (len_0, len_1, ...))."""
        lens = [int(edc_len)] * self.num_bands if np.isscalar(edc_len) else [int(v) for v in edc_len]
        if len(lens) != self.num_bands:
            raise ValueError("precompute_decay_targets: one EDC window length per band")
        key = (edc_start, lens[0] if len(set(lens)) == 1 else tuple(lens))
        if (self.edr_store is not None and self.edc_store is not None and self.edr_store[0] == win
                and self.edc_store[0] == key):
            return                      # the stacked stores already hold these targets
        Lmax = max(lens)
        edr_T, edr_s, edc_T = [], [], []
        for d, L in zip(self.datasets, lens):
            if (d.edr_store is None or d.edc_store is None or d.edr_store[0] != win
                    or d.edc_store[0] != (edc_start, L)):
                d.precompute_decay_targets(win, edc_start, L, chunk)
            edr_T.append(d.edr_store[1])
            edr_s.append(d.edr_store[2])
            T = d.edc_store[1]
            edc_T.append(T if L == Lmax else torch.nn.functional.pad(T, (0, Lmax - L)))
        self.edr_store = (win, torch.cat(edr_T), torch.cat(edr_s))
        self.edc_store = (key, torch.cat(edc_T).contiguous())
        for d in self.datasets:                  # the stacked copies are the live ones
            d.edr_store = d.edc_store = None

    def slot_grid(self, bins: torch.Tensor, conj: torch.Tensor) -> torch.Tensor:
        """z_s (1 + S,) complex128: the frequency grid on the slot order of ``ops.irfft_slot_order`` -- column 0 = bin 0,
        column 1 + s = bin bins[s], conjugated where conj[s].  Built once, cached."""
        if getattr(self, '_slot_z', None) is None:
            z = self.z_values
            self._slot_z = torch.cat([z[:1], torch.where(conj, z[bins].conj(), z[bins])])
        return self._slot_z

    def slot_ordered(self, bins: torch.Tensor, conj: torch.Tensor):
        """(z_s (1 + S,) complex128, early_s (bands*R, 1 + S) complex64): the frequency grid and the early-response
        store on the slot-ordered grid (a real signal's spectrum at conj(z) is the conjugate).  Built once, cached."""
        if getattr(self, '_slot_cache', None) is None:
            zs = self.slot_grid(bins, conj)
            E = self.early_rir_mag_response
            out = torch.empty((E.shape[0], 1 + bins.numel()), dtype=E.dtype, device=E.device)
            out[:, 0] = E[:, 0]
            for r0 in range(0, E.shape[0], 512):                      # bounded temporaries
                blk = E[r0:r0 + 512][:, bins]
                out[r0:r0 + 512, 1:] = torch.where(conj, blk.conj(), blk)
            self._slot_cache = (zs, out)
        return self._slot_cache

    # the direct-path store of the linear step built with float64 transforms (csrc/fft64.hip); False: the float32 transforms
    # of csrc/fft.hip (rounds 4-5; the cross-check of tests/test_gpu_round6.py)
    direct_time_f64 = True

    def direct_time(self, filt: Optional[torch.Tensor], n: int, chunk: int = 128) -> torch.Tensor:
        """xd (bands*R, n) float32: every receiver's direct path through its band's filter in the TIME domain,
        xd[band R + r] = irfft(early[r] filt_band, n) -- the part of x = irfft((sum_g gain_g T_g + d) filt, n) (reference
        model.py:619, trainer.py:459, losses.py:207-213 / :442-445) that no parameter touches.  The transform is linear,
        so the step adds the band's G transformed group responses to these rows instead of transforming every receiver's
        spectrum (csrc/linear.hip).  A constant of the dataset like the decay targets: built once, cached per filter."""
        key = (None if filt is None else (filt.data_ptr(), tuple(filt.shape)), int(n), bool(self.direct_time_f64))
        cache = getattr(self, '_direct_time', None)
        if cache is None or cache[0] != key:
            E = self.early_rir_mag_response
            R, Ku = self.R, (n + 1) // 2
            if E.shape[0] != self.num_bands * R or E.shape[1] < Ku:
                raise RuntimeError("direct_time: the early-response store does not match bands x receivers")
            out = torch.empty((E.shape[0], n), dtype=torch.float32, device=E.device)
            # Round 6: the store in FLOAT64, rounded to float32 once.  The reference transforms complex128 spectra
            # (losses.py:442-445, model.py:618-619) of a float64 rfft (dataloader.py:250); a float32 transform leaves 3e-7 of a
            # row's largest sample on every sample, which is percent-level on the last tenth of the EDC window and reached
            # dL/dM through the dB stages.  Built once per dataset: csrc/fft64.hip, speed irrelevant.
            f64 = self.direct_time_f64 and n >= 3 and (n & 1)
            Et = self.early_rir_time if f64 else None
            nfft = 2 * (E.shape[1] - 1)
            if Et is not None and (nfft < 2 or nfft & (nfft - 1) or Et.shape[0] != E.shape[0] or Et.shape[1] > nfft):
                Et = None                 # (no time-domain store that matches: the complex64 spectra, transformed in float64)
            for q in range(self.num_bands):
                if f64:
                    f = None if filt is None else filt[q, :Ku].to(torch.complex128).contiguous()
                    for r0 in range(q * R, (q + 1) * R, 32):
                        r1 = min(r0 + 32, (q + 1) * R)
                        X = ops.rfft_pow2_f64(Et[r0:r1], nfft, Ku) if Et is not None else E[r0:r1, :Ku].to(torch.complex128)
                        out[r0:r1] = ops.irfft_odd_f64(X, n, filt=f)
                    continue
                f = None if filt is None else filt[q, :Ku].to(torch.complex64)
                for r0 in range(q * R, (q + 1) * R, chunk):
                    r1 = min(r0 + chunk, (q + 1) * R)
                    H = E[r0:r1, :Ku].to(torch.complex64)
                    out[r0:r1] = ops.irfft_odd_fwd((H * f) if f is not None else H.contiguous(), n)
            self._direct_time = (key, out, filt)          # (the entry keeps the filter alive: the key is its ADDRESS, and a
                                                          # freed filter's block handed to another one would be a false hit)
        return self._direct_time[1]

    def edr_target_tiled(self) -> torch.Tensor:
        """The target EDR store in the tiled cell order of csrc/edrlin.hip (``ops.spec_tile``); built once, cached."""
        T = self.edr_store[1]
        cache = getattr(self, '_edr_tiled', None)
        if cache is None or cache[0] != (T.data_ptr(), tuple(T.shape)):
            out = torch.empty_like(T)
            for r0 in range(0, T.shape[0], 256):
                out[r0:r0 + 256] = ops.spec_tile(T[r0:r0 + 256])
            self._edr_tiled = ((T.data_ptr(), tuple(T.shape)), out, T)
        return self._edr_tiled[1]

    def direct_stft(self, filt: Optional[torch.Tensor], n: int, win: int, chunk: int = 64, tiled: bool = False) -> torch.Tensor:
        """Sd (bands*R, nframes, win / 2 + 1) complex64: the STFT (Hann ``win``, hop win / 2, as losses.py:501-553) of every
        row of ``direct_time`` -- the short-time spectrum of a receiver's signal is Sd[row] + sum_g gain_g STFT(tau_g) (the
        STFT is linear), which is what the EDR kernels of csrc/edrlin.hip compose on the fly.  Built once, cached."""
        key = (None if filt is None else (filt.data_ptr(), tuple(filt.shape)), int(n), int(win), bool(tiled))
        cache = getattr(self, '_direct_stft', None)
        if cache is None or cache[0] != key:
            xd = self.direct_time(filt, n)
            rows = xd.shape[0]
            out = torch.empty((rows, ops.stft_nframes(n, win), win // 2 + 1), dtype=torch.complex64, device=xd.device)
            for r0 in range(0, rows, chunk):
                blk = xd[r0:r0 + chunk]
                m = blk.shape[0]
                if m % 2:
                    blk = torch.cat([blk, torch.zeros_like(blk[:1])])
                x2 = torch.stack((blk[0::2], blk[1::2]), dim=-1).contiguous()
                out[r0:r0 + m] = ops.stft_pairs_spectrum(x2, m, win, tiled=tiled)
            self._direct_stft = (key, out, filt)
        return self._direct_stft[1]

    def global_rows(self, per_band: Sequence[Sequence[int]]) -> List[int]:
        """per_band[q] = receiver indices of band q's batch -> band-major global rows."""
        if len(per_band) != self.num_bands or len({len(s) for s in per_band}) != 1:
            raise ValueError("global_rows: one equally sized index list per band")
        return [q * self.R + int(i) for q, sel in enumerate(per_band) for i in sel]

    def collate(self, rows, lean="rows") -> Dict:
        if lean != "rows":
            raise ValueError('BandStackedDataset serves lean="rows" batches only')
        if self.edr_store is None or self.edc_store is None:
            raise RuntimeError('collate needs precompute_decay_targets() first')
        idx = rows if torch.is_tensor(rows) else torch.as_tensor(list(rows), dtype=torch.long, device=self.device)
        return {'z_values': self.z_values, 'norm_listener_position': self.norm_listener_position,
                'target_early_response': self.early_rir_mag_response, 'edr_target': self.edr_store,
                'edc_target': self.edc_store, 'receiver_index': idx, 'row_index': idx, 'dataset': self}


def bank_shards(orders: Sequence[Sequence[int]], batch: int, rank: int = 0, world_size: int = 1):
    """Steps of one pass over the bands' receiver orders (equal lengths): yields, per step, THIS RANK's receivers of
    every band -- an equal share of the band's (global) batch of ``batch`` receivers (dataloader.rank_shard), the same
    number of steps on every rank, a ragged tail smaller than the world skipped everywhere."""
    from .dataloader import rank_shard
    n = len(orders[0])
    for i0 in range(0, n, batch):
        if min(batch, n - i0) // world_size == 0:
            continue
        yield [rank_shard(o[i0:i0 + batch], rank, world_size) for o in orders]


class BandFlatAdam(FlatAdam):
    """FlatAdam whose learning-rate table is per (group, band): a band that has stopped early
    (trainer.py:410-418) keeps stepping with lr = 0, i.e. its parameters no longer move."""

    def __init__(self, groups, num_bands: int, extra_slots: int = 0):
        self.num_bands = num_bands
        self.band_active = [True] * num_bands
        super().__init__(groups, extra_slots=extra_slots)
        ng = len(self.param_groups)
        if ng * num_bands > 255:
            raise ValueError("too many (group, band) learning-rate segments")
        # every leaf is (bands, ...) contiguous: band q owns the q-th equal chunk of its flat range
        seg = self.seg.cpu()
        off = 0
        for gi, g in enumerate(self.param_groups):
            for p in g['params']:
                k = p.numel() // num_bands
                for q in range(num_bands):
                    seg[off + q * k: off + (q + 1) * k] = gi * num_bands + q
                off += p.numel()
        self.seg = seg.to(self.flat_param.device)
        self.lr_seg = torch.zeros(ng * num_bands, dtype=torch.float32, device=self.flat_param.device)
        self._lr_host = None
        self.sync_lr()

    def sync_lr(self):
        if not hasattr(self, 'band_active') or self.lr_seg.numel() != len(self.param_groups) * self.num_bands:
            return            # called from the base constructor before the table exists
        lrs = [float(g['lr']) * (1.0 if self.band_active[q] else 0.0)
               for g in self.param_groups for q in range(self.num_bands)]
        if lrs != self._lr_host:
            self.lr_seg.copy_(torch.tensor(lrs, dtype=torch.float32), non_blocking=False)
            self._lr_host = lrs


class BandBankTrainer:
    """VarReceiverPosTrainer (trainer.py:338-564) for a BandBank: the same step, all bands at once.

    ``subband_filter_freq_resp``: (bands, K) complex responses of the bands' sub-band filters
    (trainer.py:112-150 derives them from pyfar FIR taps; they are input data here).
    Losses are (bands,) vectors; band q's entries equal what band q's own trainer reports."""

    capturable = True
    concurrent_branches = True
    use_slot_order = True        # evaluate the main branch on the irfft's slot-ordered grid when the length has one
    use_pairs = True             # ... and carry two items per transform, time signals pair-interleaved
    use_fused = True             # blocks of <= 4 lines: the explicit launch sequence on the polynomial form (bankstep.py)

    def __init__(self, bank: BandBank, trainer_config: TrainerConfig,
                 subband_filter_freq_resp: Optional[torch.Tensor] = None, process_group=None,
                 stft_win: int = 4096, band_names: Optional[Sequence] = None, data_parallel: bool = True):
        """``data_parallel=False``: this process trains its bands on its own even when torch.distributed is initialised --
        the band-sharded placement (whole bands over ranks, run_subband_training_treble.py:175-204 spread over GPUs), which
        needs no collective at all."""
        cfg = trainer_config
        if not cfg.use_colorless_loss or not bank.use_colorless_loss:
            raise NotImplementedError("BandBankTrainer follows the sub-band recipe (colorless loss on)")
        if cfg.use_reg_loss or cfg.use_erb_edr_loss or cfg.use_frequency_weighting or cfg.reduced_pole_radius != 1.0:
            raise NotImplementedError("BandBankTrainer: plain EDR + EDC + colorless losses on the unit circle")
        self.net, self.config = bank, cfg
        self.num_bands = bank.num_bands
        self.band_names = list(band_names) if band_names is not None else list(range(self.num_bands))
        self.max_epochs, self.patience = cfg.max_epochs, 5
        self.train_dir = cfg.train_dir
        self.stft_win = stft_win
        self.subband_filter_freq_resp = None
        if cfg.subband_process_config is not None:
            if subband_filter_freq_resp is None or subband_filter_freq_resp.shape[0] != self.num_bands:
                raise ValueError("pass subband_filter_freq_resp as (bands, K)")
            self.subband_filter_freq_resp = subband_filter_freq_resp.to(torch.complex64).contiguous()
        self._filt_u = None
        # Adam groups by the reference's name rules (init_scheduler :152-228)
        # (the gain network last: its range of the flat buffers can then be stepped on its own, bankstep.py)
        groups = [{'params': [bank.output_gains], 'lr': cfg.io_lr},
                  {'params': [bank.input_gains], 'lr': cfg.io_lr},
                  {'params': [bank.feedback_loop_M], 'lr': cfg.lr},
                  {'params': [bank.output_scalars_w], 'lr': cfg.io_lr}]
        # (3 loss slots per band ride the gradient bucket of a data-parallel job: EDR, EDC, colorless share)
        self.optimizer = BandFlatAdam(groups, self.num_bands, extra_slots=3 * self.num_bands)
        bank.relink()                               # the leaves now live in the flat buffer
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=10, gamma=0.1)
        # every band's EDC window ends at ITS longest decay time (trainer.py:56-59: max_ir_len_ms = T60max of the band's
        # decay times; run_subband_training_treble.py:286 gives every band its own dataset's).  The bank's criterion
        # carries the longest of them; bands with shorter windows are handled by the banded EDC kernels (per-item
        # window lengths, per-band mask rows: _band_windows / _item_windows)
        self.band_ir_len_ms = [float(np.max(np.asarray(net.common_decay_times))) * 1e3
                               if net.common_decay_times is not None else 2000.0 for net in bank.nets]
        self.max_ir_len_ms = max(self.band_ir_len_ms)
        self.criterion = [None, edc_loss(self.max_ir_len_ms, bank.sample_rate, use_mask=cfg.use_edc_mask)]
        self._band_criteria = [edc_loss(ms, bank.sample_rate, use_mask=cfg.use_edc_mask) for ms in self.band_ir_len_ms]
        self._win_cache = {}
        self.process_group = process_group
        sharded = data_parallel and dist.is_initialized()
        self.world_size = dist.get_world_size(process_group) if sharded else 1
        self.rank = dist.get_rank(process_group) if sharded else 0
        self._allreduce = None
        if self.world_size > 1:
            opt, pg = self.optimizer, process_group
            # ONE collective per step: gradients of all bands + the loss terms of all bands
            self._allreduce = lambda: dist.all_reduce(opt.bucket, op=dist.ReduceOp.SUM, group=pg)
        self.allreduce_in_graph = True      # capture the all-reduce inside the step's HIP graph (RCCL is capturable)
        self._side = self._side2 = None
        self._fused = FusedBankStep(self) if (self.use_fused and FusedBankStep.supported(self)) else None
        if self.use_fused and self._fused is None and self.rank == 0:
            import warnings
            warnings.warn("BandBankTrainer: this layout (more than 8 lines per group -- more than 4 off the unit circle --, "
                          "more than 4 groups per band or more than 64 blocks in the bank) is outside the explicit "
                          "block-transfer-function step; the bank "
                          "steps through the per-bin elimination kernels under autograd (same results, slower)",
                          RuntimeWarning, stacklevel=2)
        # leaves are first touched on one stream and receive gradients from the other by design (§5.1): the
        # engine synchronises them; its per-backward warning about that would only hide real messages
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

    def _decay_window(self, K: int):
        """(start, length of the LONGEST band window) -- the extent the time-domain buffers and mask rows are sized for"""
        return self.criterion[1].window(K)

    def _band_windows(self, K: int) -> Optional[List[int]]:
        """Per-band EDC window lengths when they differ (None: one window for all bands)."""
        lens = [c.window(K)[1] for c in self._band_criteria]
        return None if len(set(lens)) == 1 else lens

    def _target_window(self, K: int):
        """(start, edc_len) for ``precompute_decay_targets``: edc_len is a list when the bands' windows differ."""
        start, length = self._decay_window(K)
        lens = self._band_windows(K)
        return start, (length if lens is None else lens)

    def _item_windows(self, K: int, Bper: int, device):
        """(item_len (bands * Bper,) int32, band_len (bands,) int32) on the device for the banded EDC kernels, or
        (None, None) when all bands share one window; cached (the captured graph holds their pointers)."""
        lens = self._band_windows(K)
        if lens is None:
            return None, None
        key = (K, Bper, str(device))
        if key not in self._win_cache:
            band_len = torch.tensor(lens, dtype=torch.int32, device=device)
            self._win_cache[key] = (band_len.repeat_interleave(Bper).contiguous(), band_len)
        return self._win_cache[key]

    def _band_mask_rows(self, keep: Optional[torch.Tensor], K: int, global_batch: int, device) -> torch.Tensor:
        """(bands, Lmax) pre-normalised EDC weights of bands with different windows: band q keeps the first len_q entries of
        the 0/1 mask ``keep`` (None: every index) and divides by (global batch x kept among them); zero behind its window."""
        _, Lmax = self._decay_window(K)
        lens = self._band_windows(K)
        k = torch.ones(Lmax, dtype=torch.float32, device=device) if keep is None else keep.to(device=device, dtype=torch.float32)
        rows = torch.zeros((self.num_bands, Lmax), dtype=torch.float32, device=device)
        for q, L in enumerate(lens):
            cnt = k[:L].sum()
            rows[q, :L] = torch.where(cnt > 0, k[:L] / (cnt * global_batch), torch.zeros_like(k[:L]))
        return rows

    @property
    def _reduced_step_losses(self) -> bool:
        """True when a training step's reported decay losses are already the whole job's (the explicit step
        all-reduces them in the gradient bucket)."""
        return self._fused is not None and self.world_size > 1

    def _stream(self, which: str):
        if not self.concurrent_branches or which in getattr(self, '_disabled_streams', ()):
            return None
        if getattr(self, which) is None:
            setattr(self, which, torch.cuda.Stream())
        return getattr(self, which)

    @torch.no_grad()
    def normalize(self, data: Dict):
        self.net.normalize(data['z_values'])
        if self._fused is not None:
            self._fused.invalidate_records()        # (b, c rescaled in place: the explicit step's kept records are stale)

    def _filter_on(self, Ku: int, order) -> Optional[torch.Tensor]:
        """(bands, Ku) sub-band filter responses on the bins the main branch is evaluated on (slot order when the
        irfft length has one); built once."""
        if self.subband_filter_freq_resp is None:
            return None
        if self._filt_u is None or self._filt_u[0] != (Ku, order is not None):
            F = self.subband_filter_freq_resp
            if order is not None:
                bins, conj = order
                Fu = torch.cat([F[:, :1], torch.where(conj, F[:, bins].conj(), F[:, bins])], dim=1)
            else:
                Fu = F[:, :Ku]
            self._filt_u = ((Ku, order is not None), Fu.contiguous())
        return self._filt_u[1]

    def _draw_edc_mask(self, length: int, Bper: int, device, draw_mask: bool = True, K: Optional[int] = None):
        """(maskw, 1 / (global batch x kept indices)) drawn like the reference (losses.py:221-227), the same on all
        ranks."""
        maskw, count = (self.criterion[1].draw_mask(length, device) if draw_mask else (None, float(length)))
        if maskw is not None and self.world_size > 1:
            dist.broadcast(maskw, src=0, group=self.process_group)
            count = float(maskw.sum().item())
        if K is not None and self._band_windows(K) is not None:
            # bands with different windows: ONE draw of the longest window, truncated per band; the rows come back
            # pre-normalised (every band has its own count), so nothing is left to divide by
            return self._band_mask_rows(maskw, K, Bper * self.world_size, device), 1.0
        return maskw, 1.0 / (Bper * self.world_size * count)

    def _fused_step(self, data: Dict, train: bool, draw_mask: bool = True):
        rows = data['row_index']
        K = data['z_values'].shape[-1]
        _, length = self._decay_window(K)
        maskw, inv = self._draw_edc_mask(length, rows.numel() // self.num_bands, rows.device, draw_mask, K=K)
        losses = self._fused.run(data, maskw, inv, normalize_first=False, train=train, allreduce=self._allreduce)
        losses.pop('_total')
        return sum(losses.values()), losses

    def _step_losses(self, data: Dict, draw_mask: bool = True, mask_prenorm: Optional[torch.Tensor] = None,
                     normalize_first: bool = False, defer_total: bool = False) -> Dict:
        """Forward + losses of one band-major batch (rows = data['row_index'], bands x B items);
        stream structure as VarReceiverPosTrainer._step_losses."""
        bank, cfg, nb = self.net, self.config, self.num_bands
        if normalize_first and self._fused is not None:
            self._fused.invalidate_records()        # (this path rescales b, c in place too)
        z = data['z_values']
        rows = data['row_index']
        Btot = rows.numel()
        if Btot % nb:
            raise ValueError("the batch must hold the same number of receivers for every band")
        Bper = Btot // nb
        main = torch.cuda.current_stream()
        side = self._stream('_side')
        fused = bank.num_delay_lines_per_group <= 4 and nb * bank.num_groups <= 64
        if fused:
            # the sub-FDN solve (+ rescale of b, c) leads the MAIN stream: everything of the main branch hangs
            # on it; the rotations and the gain network lead the side stream, followed there by the tail of
            # the colorless node (statistics, loss terms, adjoint kernel)
            rgain = None
            if side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    Q, QQ = bank.rotations()
                    rot_done = torch.cuda.Event()
                    rot_done.record(side)
                    rgain = bank.group_gains(data['norm_listener_position'], rows)
                    mlp_done = torch.cuda.Event()
                    mlp_done.record(side)
            else:
                Q, QQ = bank.rotations()
            gridK = FrequencyGrid.of(z)
            with torch.no_grad():
                Ys, Ss, en = ops.subfdn_colorless_fwd(gridK.turns, gridK.logr, bank._blocks(), bank.delays,
                                                      bank.input_gains.data.view(-1), bank.output_gains.data.view(-1),
                                                      normalize_first)
            if side is not None:
                sub_done = torch.cuda.Event()
                sub_done.record(main)
        else:
            if side is not None:
                side.wait_stream(main)
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                if normalize_first:
                    self.normalize(data)
                Q, QQ = bank.rotations()
                if side is not None:
                    ready = torch.cuda.Event()
                    ready.record(side)
                S = bank.sub_fdn_group_sums(z)
                extra, spec, sparse = ColorlessTerms.apply(S, Q, cfg.use_asym_spectral_loss,
                                                           cfg.spectral_loss_weight, cfg.sparsity_loss_weight,
                                                           shard_loss_scales(self.world_size, 1, 1.0)['colorless'], True, nb)
        # a tensor allocated on one stream and read on the other must be recorded there: otherwise its block
        # returns to the allocating stream's pool the moment autograd drops it and a kernel of that stream may
        # overwrite it while the other stream still reads it (seen as wrong dL/dM, dL/db under graph replay)
        if fused and side is not None:
            main.wait_event(rot_done)
            QQ.record_stream(main)
        elif fused:
            rgain = bank.group_gains(data['norm_listener_position'], rows)
        else:
            rgain = bank.group_gains(data['norm_listener_position'], rows)
            if side is not None:
                main.wait_event(ready)
                QQ.record_stream(main)
        K = z.shape[-1]
        Ku = (K + 1) // 2 if K % 2 == 1 else K          # irfft(X, n = K) reads bins 0..(K-1)/2 only
        # Slot order (n = 65 537): the irfft's natural input order is a fixed permutation of the bins, some of them
        # conjugated; solve and output stage are pointwise in the bin, so they are evaluated directly on the
        # permuted grid { z_k or conj(z_k) } against the (once) permuted early-response store and filter, and
        # the transform needs neither the gather in front nor the scatter behind its adjoint.
        order = ops.irfft_slot_order(K, z.device) if (self.use_slot_order and 'dataset' in data) else None
        if order is not None:
            zu, direct = data['dataset'].slot_ordered(*order)
        else:
            zu, direct = z[:Ku], data['target_early_response'][:, :Ku]
        Y = bank.delay_line_responses(zu, QQ)
        filt = self._filter_on(Ku, order)
        if fused and side is not None:
            main.wait_event(mlp_done)
            rgain.record_stream(main)
        H = OutputStage.apply(Y, bank.output_gains.view(-1), rgain, bank.num_delay_lines_per_group,
                              direct, filt, rows, nb)
        if fused:
            # the loss side of the colorless branch is issued AFTER the main solve / output stage (the graph
            # executor launches nodes in capture order: its small kernels would otherwise sit in front of them)
            if side is not None:
                side.wait_event(sub_done)
                for t in (Ys, Ss, en):
                    t.record_stream(side)
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                extra, spec, sparse = SubFdnColorless.apply(
                    bank._blocks(), bank.input_gains.view(-1), bank.output_gains.view(-1), Q, Ys, Ss, en, gridK,
                    bank.delays, normalize_first, cfg.use_asym_spectral_loss, cfg.spectral_loss_weight,
                    cfg.sparsity_loss_weight, shard_loss_scales(self.world_size, 1, 1.0)['colorless'], nb,
                    torch.is_grad_enabled())
                if side is not None:
                    tail_done = torch.cuda.Event()
                    tail_done.record(side)
        start, length = self._decay_window(K)
        item_len, _ = self._item_windows(K, Bper, H.device)
        gb = Bper
        if mask_prenorm is not None:
            maskw, count = mask_prenorm, None
        elif item_len is not None:
            mask_prenorm, _ = self._draw_edc_mask(length, Bper, H.device, draw_mask, K=K)     # (bands, Lmax) rows
            maskw, count = mask_prenorm, None
        else:
            maskw, count = (self.criterion[1].draw_mask(length, H.device) if draw_mask else (None, float(length)))
            if maskw is not None and self.world_size > 1:
                dist.broadcast(maskw, src=0, group=self.process_group)
                count = float(maskw.sum().item())
            gb = Bper * self.world_size
        edr_t, edc_t = data['edr_target'], data['edc_target']
        pairs_path = order is not None and self.use_pairs and self.stft_win == 4096
        total, edr_v, edc_v = decay_losses(
            H, None, win=self.stft_win, edr_weight=cfg.edr_loss_weight, edc_weight=cfg.edc_loss_weight,
            edc_start=start, edc_len=length, edc_maskw=maskw, edc_count=count,
            edc_maskw_prenormalised=mask_prenorm is not None, global_batch=gb,
            edr_target=(edr_t[1], edr_t[2]), edc_target=edc_t[1], side_stream=self._stream('_side2'),
            unit_grad=True, n_time=K, target_rows=rows, nbands=nb, slot_order=order is not None,
            pairs=pairs_path, edc_item_len=item_len, edc_items_per_band=Bper,
            join_event=tail_done if (fused and side is not None) else None)
        losses = {'edc_loss': edc_v, 'edr_loss': edr_v, 'spectral_loss': spec.detach(),
                  'sparsity_loss': sparse.detach()}
        if side is not None:
            if not (fused and pairs_path):        # (the pair path already joined the side branch behind the transform)
                main.wait_stream(side)
            extra.record_stream(main)
        if defer_total:
            losses['_heads'] = [total, extra]
        else:
            losses['_total'] = total + extra
        return losses

    def _ones(self, dev):
        if getattr(self, '_unit', None) is None:
            self._unit = torch.ones(() if self.num_bands == 1 else (self.num_bands,), dtype=torch.float32,
                                    device=dev)
        return self._unit

    def train_step(self, data: Dict):
        """One optimiser step of every band (trainer.py:452-477); returns ((bands,) totals, parts)."""
        if self._fused is not None:
            return self._fused_step(data, train=True)
        self.optimizer.zero_grad(set_to_none=True)
        losses = self._step_losses(data, defer_total=True)
        heads = losses.pop('_heads')
        one = self._ones(heads[0].device)
        torch.autograd.backward(heads, [one] * len(heads))
        self.optimizer.pack_grads()
        if self._allreduce is not None:
            self._allreduce()
        self.optimizer.step()
        return sum(losses.values()), losses

    @torch.no_grad()
    def valid_step(self, data: Dict):
        if self._fused is not None:
            return self._fused_step(data, train=False)
        losses = self._step_losses(data)
        losses.pop('_total')
        return sum(losses.values()), losses

    def graphed(self, dataset: BandStackedDataset, batch_per_band: int, mask_source: str = "device",
                mask_seed: Optional[int] = None):
        """normalize + train_step of bands x batch_per_band receivers as one HIP-graph replay."""
        from .trainer import GraphedTrainStep
        return GraphedTrainStep(self, dataset, batch_per_band * self.num_bands, mask_source, mask_seed)

    def save_model(self, e: int):
        """checkpoints/model_e{e}.pt per band under train_dir/band_<name>/ (save_model :249-257)."""
        if self.rank != 0:
            return
        for q, name in enumerate(self.band_names):
            d = os.path.join(self.train_dir, f'band_{name}', 'checkpoints')
            os.makedirs(d, exist_ok=True)
            torch.save(self.net.band_state_dict(q), os.path.join(d, 'model_e' + str(e) + '.pt'))

    def train(self, dataset: BandStackedDataset, train_indices: Sequence[Sequence[int]],
              valid_indices: Sequence[Sequence[int]], batch_size: Optional[int] = None,
              save_checkpoints: bool = True, log: bool = True):
        """Epoch loop of every band in lockstep (trainer.py:345-424): per band its own shuffled
        receiver order, normalize + train_step per batch, validation, StepLR, early stopping
        (a stopped band's learning rates drop to 0; the loop ends when every band has stopped).
        ``train_indices[q]`` / ``valid_indices[q]``: band q's receiver indices (load_dataset's split)."""
        nb = self.num_bands
        B = batch_size or self.config.batch_size
        ntr = len(train_indices[0])
        if any(len(t) != ntr for t in train_indices) or len({len(v) for v in valid_indices}) != 1:
            raise ValueError("BandBankTrainer.train: the bands' splits must have equal sizes")
        world, rank = self.world_size, self.rank
        if B // world == 0:
            raise ValueError("BandBankTrainer.train: the batch is smaller than the number of ranks")
        K = dataset.z_values.shape[-1]
        dataset.precompute_decay_targets(self.stft_win, *self._target_window(K))
        Bl = B // world                       # this rank's receivers per band of a full batch (global batch B)
        step = self.graphed(dataset, Bl)
        self.train_loss = [[] for _ in range(nb)]
        self.valid_loss = [[] for _ in range(nb)]
        self.individual_train_loss, self.individual_valid_loss = [], []
        self.epoch_times = []
        early = [0] * nb
        # data-parallel: every rank must walk the SAME shuffled orders (it takes its share of every batch): one seed
        # for all ranks, drawn once; a single process keeps the global generator, as the reference does
        gen = None
        if world > 1:
            seed_t = torch.randint(0, 2 ** 62, (1,), dtype=torch.long).to(dataset.device)
            dist.broadcast(seed_t, src=0, group=self.process_group)
            gen = torch.Generator().manual_seed(int(seed_t.item()))
        st = time.time()
        if save_checkpoints:
            self.save_model(-1)
        for epoch in range(self.max_epochs):
            t0 = time.time()
            orders = [[train_indices[q][i] for i in torch.randperm(ntr, generator=gen).tolist()] for q in range(nb)]
            agg_t, nsteps = {}, 0
            shards = [dataset.global_rows(sel) for sel in bank_shards(orders, B, rank, world)]
            full = [r for r in shards if len(r) == Bl * nb]
            # the epoch's full batches go to the device as ONE schedule; every replayed step fetches the next one's
            # receivers itself (GraphedTrainStep.run_schedule)
            for cur in step.run_schedule(full):
                for k, v in cur.items():
                    if not k.startswith('_'):
                        agg_t[k] = agg_t.get(k, 0.0) + v.detach()
                nsteps += 1
            for rows in shards:
                if len(rows) == Bl * nb:
                    continue
                batch = dataset.collate(rows)                         # ragged tail: host launches
                self.normalize(batch)
                _, cur = self.train_step(batch)
                for k, v in cur.items():
                    agg_t[k] = agg_t.get(k, 0.0) + v.detach()
                nsteps += 1
            agg_v, nv = {}, 0
            for sel in bank_shards(valid_indices, B, rank, world):
                _, cur = self.valid_step(dataset.collate(dataset.global_rows(sel)))
                for k, v in cur.items():
                    agg_v[k] = agg_v.get(k, 0.0) + v.detach()
                nv += 1
            self.scheduler.step()
            if world > 1 and not self._reduced_step_losses:
                # (per-step values are rank shares unless the step itself reduced them with the gradients)
                from .trainer import reduce_epoch_losses
                agg_t = reduce_epoch_losses(agg_t, self.process_group)
            if world > 1:
                from .trainer import reduce_epoch_losses
                agg_v = reduce_epoch_losses(agg_v, self.process_group)
            tl = (sum(agg_t.values()) / max(nsteps, 1)).tolist()      # one sync per epoch
            from .losses import raise_on_unit_grad_violation
            raise_on_unit_grad_violation()                             # (the autograd path's unit-gradient promise)
            vl = (sum(agg_v.values()) / max(nv, 1)).tolist() if agg_v else [0.0] * nb
            self.individual_train_loss.append({k: (v / max(nsteps, 1)).tolist() for k, v in agg_t.items()})
            self.individual_valid_loss.append({k: (v / max(nv, 1)).tolist() for k, v in agg_v.items()})
            for q in range(nb):
                if not self.optimizer.band_active[q]:
                    continue
                self.train_loss[q].append(tl[q])
                self.valid_loss[q].append(vl[q])
                if epoch >= 1:
                    early[q] = early[q] + 1 if abs(self.valid_loss[q][-2] - self.valid_loss[q][-1]) <= 1e-3 else 0
                if early[q] == self.patience:
                    self.optimizer.band_active[q] = False
            self.optimizer.sync_lr()
            if save_checkpoints:
                self.save_model(epoch)
            self.epoch_times.append(time.time() - t0)
            if log and self.rank == 0:
                print(f"epoch {epoch}: " + " ".join(f"[{self.band_names[q]}] {tl[q]:.3f}/{vl[q]:.3f}"
                                                    for q in range(nb)) + f" ({time.time() - t0:.2f} s)")
            if not any(self.optimizer.band_active):
                break
        self.train_time = time.time() - st
