"""Trainer of the differentiable GFDN on the MI355X hot path.

Counterpart of the reference's src/diff_gfdn/trainer.py: Trainer (:26-332) and
VarReceiverPosTrainer (:338-564) with identical step semantics --
  * loss = w_edr EDR + w_edc EDC + sum_g w_spec (a)MSE(Hout[:,g], 1) + w_sp sparsity(Q_last)
    (calculate_losses :259-315; the sparsity term is ASSIGNED per group, so only the last group
    counts -- reproduced);
  * normalize (:317-332): before every step, b_n, c_n /= (mean_k |Hout[k,g]|^2)^(1/4);
  * Adam with per-name learning-rate groups and StepLR(10, 0.1) (init_scheduler :152-228);
  * checkpoints model_e{e}.pt of the state dict (save_model :249-257).
The step itself runs the fused path: one per-bin solve shared by the batch, the output stage
with the sub-band filter folded in, ONE irfft of H feeding both decay losses, one adjoint.

Multi-GPU: one process per GPU; a batch of receiver positions is sharded over the
ranks; after backward every rank contributes its gradients to ONE flat fp32 all-reduce (RCCL
over xGMI).  Position-independent terms (colorless + sparsity) are pre-divided by the world
size so that the sum equals the single-process gradient.
"""
import contextlib
import os
import time
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .colorless_losses import amse_loss, group_spectral_loss, mse_loss, sparsity_loss
from .config import CouplingMatrixType, TrainerConfig
from .functional import ColorlessTerms, OutputStage, SHToDirectional, irfft_like_torch
from . import hip_ops as ops
from .hip_ops import normalize_io, spectral_stats
from .losses import decay_losses, directional_edc_loss, edc_loss, edr_loss, ms_to_samps, shard_loss_scales
from .model import DiffGFDN
from .optim import FlatAdam


@torch.no_grad()
def get_response(x, net, output_scalars=None):
    """(H, [H_sub_fdn,] h) with h = irfft(H) at the default length n = 2(K-1)
    (reference utils.py:149-179)."""
    if getattr(net, 'use_colorless_loss', False):
        H, H_sub = net(x, output_scalars) if output_scalars is not None else net(x)
        return H, H_sub, irfft_like_torch(H)
    H = net(x)
    return H, irfft_like_torch(H)


def ir_filenames(filename_prefix: str, src_pos: torch.Tensor, rec_pos: torch.Tensor):
    """One file name per receiver as the reference writes them (trainer.py:537-556); yields (receiver, name)."""
    src_pos, rec_pos = src_pos.detach().cpu(), rec_pos.detach().cpu()
    num_src = 1 if src_pos.ndim == 1 or bool(torch.all(src_pos == src_pos[0])) else src_pos.shape[0]
    for s_ in range(num_src):
        for r in range(rec_pos.shape[0]):
            if num_src == 1:
                yield r, f'{filename_prefix}_({rec_pos[r, 0]:.2f}, {rec_pos[r, 1]:.2f}, {rec_pos[r, 2]:.2f}).wav'
            else:
                yield r, (f'{filename_prefix}_src_pos=({src_pos[s_, 0]:.2f}, {src_pos[s_, 1]:.2f}, '
                          f'{src_pos[s_, 2]:.2f})_rec_pos=({rec_pos[r, 0]:.2f}, {rec_pos[r, 1]:.2f}, '
                          f'{rec_pos[r, 2]:.2f}).wav')


def write_wav(path: str, sample_rate: int, samples: np.ndarray):
    """32-bit float wav, samples (T, channels) -- the host side of the reference's torchaudio.save calls."""
    from scipy.io import wavfile
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    wavfile.write(path, int(sample_rate), np.ascontiguousarray(samples, dtype=np.float32))


class FlatGradAllReduce:
    """One flat fp32 buffer for all parameter gradients -> a single all-reduce per step."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)

    def __call__(self):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n


def reduce_epoch_losses(agg: Dict[str, torch.Tensor], group=None) -> Dict[str, torch.Tensor]:
    """Per-epoch loss aggregates of a data-parallel job -> the values of the WHOLE job, identical on every rank
    (one all-reduce of one small tensor): the decay terms are per-rank shares of the global loss (EDR: sum over the
    rank's items, EDC: the rank's share of the global mean) and add up; the colorless terms are the same on every
    rank and are averaged.  The early-stopping test (reference trainer.py:418-424) must run on these -- a decision
    taken from rank-local values can differ between ranks, and the rank that leaves the epoch loop alone leaves the
    others waiting in the next all-reduce."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or not agg:
        return agg
    keys = sorted(agg)
    world = dist.get_world_size(group)
    vals = [torch.as_tensor(agg[k], dtype=torch.float32).reshape(-1) for k in keys]
    sizes = [v.numel() for v in vals]
    dev = next((v.device for v in vals if v.is_cuda), vals[0].device)
    flat = torch.cat([v.to(dev) for v in vals])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    out, off = {}, 0
    for k, v, n in zip(keys, vals, sizes):
        t = flat[off:off + n].reshape(torch.as_tensor(agg[k]).shape)
        out[k] = t if k in ('edr_loss', 'edc_loss') else t / world
        off += n
    return out


class Trainer:
    """Parent class (reference trainer.py:26-332)."""

    def __init__(self, net: DiffGFDN, trainer_config: TrainerConfig,
                 subband_filter_freq_resp: Optional[torch.Tensor] = None,
                 process_group=None, stft_win: int = 4096, capturable: bool = False):
        self.net = net
        if hasattr(net, 'per_delay_output'):
            net.per_delay_output = False          # (no loss reads Hout_per_del; see DiffGFDN.per_delay_output)
        self.device = trainer_config.device
        self.max_epochs = trainer_config.max_epochs
        self.patience = 5
        self.early_stop = 0
        self.train_dir = Path(trainer_config.train_dir).resolve()
        self.ir_dir = Path(trainer_config.ir_dir).resolve()
        self.use_reg_loss = trainer_config.use_reg_loss
        if self.use_reg_loss:
            raise NotImplementedError("reg_loss needs the SVF output filters (SURVEY §8 f-2)")
        self.use_colorless_loss = trainer_config.use_colorless_loss
        self.reduced_pole_radius = trainer_config.reduced_pole_radius
        self.subband_process_config = trainer_config.subband_process_config
        self.use_directional_fdn = getattr(net, 'ambi_order', None) is not None
        # the reference derives the sub-band filter from pyfar FIR taps (trainer.py:112-150);
        # pyfar is not part of this build: the (K,) response is an input
        self.subband_filter_freq_resp = subband_filter_freq_resp
        if self.subband_process_config is not None and subband_filter_freq_resp is None:
            raise ValueError("subband_process_config set: pass subband_filter_freq_resp (K,) explicitly")
        self.config = trainer_config
        self.stft_win = stft_win
        self.capturable = capturable
        self.init_scheduler(trainer_config)

        if net.common_decay_times is None:
            max_ir_len_ms = 2000
        else:
            max_ir_len_ms = float(np.max(np.asarray(net.common_decay_times))) * 1e3
        self.max_ir_len_ms = max_ir_len_ms
        if trainer_config.use_erb_edr_loss:
            # (the loss module itself takes the grouping -- losses.edr_loss(use_erb_grouping=True) -- but the fused steps of
            # these trainers evaluate the ungrouped EDR on the half grid with precomputed targets: fail loudly instead of
            # training on a different loss than the one configured)
            raise NotImplementedError("use_erb_edr_loss: the trainers' fused steps evaluate the ungrouped EDR loss")
        if self.use_directional_fdn:
            self.criterion = [directional_edc_loss(net.common_decay_times, max_ir_len_ms,
                                                   net.sample_rate,
                                                   use_mask=trainer_config.use_edc_mask)]
            self.loss_weights = torch.tensor([trainer_config.edc_loss_weight])
        else:
            self.criterion = [
                edr_loss(net.sample_rate, win_size=stft_win, hop_size=stft_win // 2,
                         reduced_pole_radius=(None if self.reduced_pole_radius == 1.0
                                              else self.reduced_pole_radius),
                         use_erb_grouping=trainer_config.use_erb_edr_loss,
                         use_weight_fn=trainer_config.use_frequency_weighting),
                edc_loss(max_ir_len_ms, net.sample_rate, use_mask=trainer_config.use_edc_mask),
            ]
            self.loss_weights = torch.tensor([trainer_config.edr_loss_weight,
                                              trainer_config.edc_loss_weight])
        if self.use_colorless_loss:
            spec = amse_loss() if trainer_config.use_asym_spectral_loss else mse_loss()
            self.colorless_criterion = [spec, sparsity_loss()]
            self.colorless_loss_weights = torch.tensor([trainer_config.spectral_loss_weight,
                                                        trainer_config.sparsity_loss_weight])
        # data-parallel state
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self._allreduce = None
        if self.world_size > 1:
            if isinstance(self.optimizer, FlatAdam):
                opt, pg = self.optimizer, process_group
                # grads are packed into the flat buffer by the caller (inside the captured graph)
                self._allreduce = lambda: dist.all_reduce(opt.flat_grad, op=dist.ReduceOp.SUM, group=pg)
            else:
                self._allreduce = FlatGradAllReduce(net.parameters(), process_group)

    # reference :152-228
    def init_scheduler(self, cfg: TrainerConfig):
        named = list(self.net.named_parameters())
        pick = lambda pred: [p for n, p in named if pred(n)]
        keys = ('feedback_loop.alpha', 'input_gains', 'output_gains', 'output_svf_params',
                'output_scalars', 'sh_output_scalars', 'input_scalars')
        groups = [
            {'params': pick(lambda n: 'feedback_loop.alpha' in n), 'lr': cfg.coupling_angle_lr},
            {'params': pick(lambda n: 'output_gains' in n), 'lr': cfg.io_lr},
            {'params': pick(lambda n: 'input_gains' in n), 'lr': cfg.io_lr},
            {'params': pick(lambda n: 'output_svf_params' in n), 'lr': cfg.io_lr},
            {'params': pick(lambda n: 'input_scalars' in n), 'lr': cfg.io_lr},
            {'params': pick(lambda n: 'output_scalars' in n or 'sh_output_scalars' in n), 'lr': cfg.io_lr},
        ]
        other = pick(lambda n: not any(k in n for k in keys))
        if other:
            groups.append({'params': other, 'lr': cfg.lr})
        if self.capturable:
            # flat-buffer fused Adam (csrc/optim.hip): step counter and learning rates live on the
            # device, so the update replays from a HIP graph; same maths as torch.optim.Adam
            self.optimizer = FlatAdam([g for g in groups if g['params']])
        else:
            self.optimizer = torch.optim.Adam(groups)
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=10, gamma=0.1)

    def save_model(self, e: int):
        if self.rank != 0:
            return
        d = os.path.join(self.train_dir, 'checkpoints')
        os.makedirs(d, exist_ok=True)
        torch.save(self.net.state_dict(), os.path.join(d, 'model_e' + str(e) + '.pt'))

    # reference :259-315 -- drop-in: separate loss modules on (H, H_sub_fdn)
    def calculate_losses(self, data: Dict, H: torch.Tensor, H_sub_fdn: Optional[Tuple] = None) -> Dict:
        if self.use_directional_fdn:
            all_losses = {'edc_loss': self.loss_weights[0].item() * self.criterion[0](
                H, data['target_common_slope_amps'])}
        else:
            tgt = data['target_rir_response']
            all_losses = {
                'edc_loss': self.loss_weights[1].item() * self.criterion[1](tgt, H),
                'edr_loss': self.loss_weights[0].item() * self.criterion[0](tgt, H),
            }
        if self.use_colorless_loss:
            spectral, sparsity = 0.0, 0.0
            fl = self.net.feedback_loop
            for k in range(self.net.num_groups):
                hk = H_sub_fdn[0][..., k]
                spectral = spectral + self.colorless_loss_weights[0].item() * self.colorless_criterion[0](
                    hk, torch.ones_like(hk))
                sparsity = self.colorless_loss_weights[1].item() * self.colorless_criterion[1](
                    fl.ortho_param(fl.M[k]))
            all_losses.update({'spectral_loss': spectral, 'sparsity_loss': sparsity})
        return all_losses

    # reference :317-332
    @torch.no_grad()
    def normalize(self, data: Dict):
        if not self.use_colorless_loss:
            return
        net = self.net
        # energy of the sub-FDN responses and the in-place rescale of b, c in two launches
        # (gfdn_subfdn_normalize); the responses themselves are never written
        from .functional import FrequencyGrid
        grid = FrequencyGrid.of(data['z_values'])
        ops.subfdn_normalize(grid.turns, grid.logr, net.feedback_loop.M.detach(), net.delay_buffer,
                             net.input_gains.data, net.output_gains.data)


    # epoch loop shared by the grid / directional trainers (reference :345-424, :697-769)
    def train(self, train_dataset, valid_dataset, save_irs: bool = False):
        self.train_loss, self.valid_loss = [], []
        self.individual_train_loss, self.individual_valid_loss = [], []
        st = time.time()
        self.save_model(-1)
        for epoch in range(self.max_epochs):
            st_epoch = time.time()
            agg_t, agg_v = {}, {}
            svf = getattr(self.net, 'use_svf_in_output', False)
            if svf:                       # full-band GFDN: b, c are normalised once per epoch (reference :365-369)
                self.normalize(next(iter(train_dataset)))
            for data in train_dataset:
                if not svf:
                    self.normalize(data)
                _, cur = self.train_step(data)
                for k, v in cur.items():
                    agg_t[k] = agg_t.get(k, 0.0) + v.detach()
            for data in valid_dataset:
                _, cur = self.valid_step(data)
                for k, v in cur.items():
                    agg_v[k] = agg_v.get(k, 0.0) + v.detach()
            self.scheduler.step()
            if isinstance(self.optimizer, FlatAdam):
                self.optimizer.sync_lr()
            nt, nv = max(len(train_dataset), 1), max(len(valid_dataset), 1)
            # data-parallel: the whole job's values, the same on every rank (the early-stop test below depends on it)
            agg_t = reduce_epoch_losses(agg_t, self.process_group)
            agg_v = reduce_epoch_losses(agg_v, self.process_group)
            agg_t = {k: float(v) / nt for k, v in agg_t.items()}     # one sync per epoch
            from .losses import raise_on_unit_grad_violation
            raise_on_unit_grad_violation()                            # (the decay losses' unit-gradient promise, checked here)
            agg_v = {k: float(v) / nv for k, v in agg_v.items()}
            self.train_loss.append(sum(agg_t.values()))
            self.valid_loss.append(sum(agg_v.values()))
            self.individual_train_loss.append(agg_t)
            self.individual_valid_loss.append(agg_v)
            self.save_model(epoch)
            if self.rank == 0:
                print(f"epoch {epoch}: train {self.train_loss[-1]:.4f} valid {self.valid_loss[-1]:.4f} "
                      f"({time.time() - st_epoch:.2f} s) " +
                      " ".join(f"{k}={v:.4f}" for k, v in agg_t.items()))
            if epoch >= 1:
                self.early_stop = self.early_stop + 1 if abs(self.valid_loss[-2] - self.valid_loss[-1]) <= 1e-3 else 0
            if self.early_stop == self.patience:
                break
        self.train_time = time.time() - st
        if save_irs:                       # reference :429-449: the trained IRs of both splits as wav files
            for split, prefix in ((train_dataset, "ir"), (valid_dataset, "valid_ir")):
                for data in split:
                    self.save_ir(data, directory=self.ir_dir, src_pos=data['source_position'],
                                 rec_pos=data['listener_position'], filename_prefix=prefix)



class VarReceiverPosTrainer(Trainer):
    """Grid-of-receivers trainer (reference trainer.py:338-564)."""

    concurrent_branches = True

    def _side_stream2(self):
        if not self.concurrent_branches or not next(self.net.parameters()).is_cuda:
            return None
        if getattr(self, '_side2', None) is None:
            self._side2 = torch.cuda.Stream()
        return self._side2

    def _side_stream(self):
        if not self.concurrent_branches or not next(self.net.parameters()).is_cuda:
            return None
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream()
        return self._side

    def _decay_window(self, K: int) -> Tuple[int, int]:
        return self.criterion[1].window(K)

    def _step_losses(self, data: Dict, draw_mask: bool = True,
                     mask_prenorm: Optional[torch.Tensor] = None, normalize_first: bool = False,
                     defer_total: bool = False) -> Dict:
        """Fused forward + losses of one batch (train_step :452-471 / valid_step :479-498).
        ``mask_prenorm``: EDC time weights already divided by (global batch x kept indices), in a
        static device buffer (graph replay); otherwise the mask is drawn here like the reference."""
        net, cfg = self.net, self.config
        if getattr(net, 'use_svf_in_output', False):
            return self._step_losses_module_forward(data, draw_mask, defer_total, mask_prenorm)
        fl = net.feedback_loop
        fl.new_forward()
        z = data['z_values']
        n = net.num_delay_lines_per_group
        # the colorless branch (sub-FDN solve -> spectral / sparsity losses) shares nothing with the
        # main branch but the parameters: fork it onto a side stream (small grids that leave most
        # of the 256 CUs idle); autograd replays each backward op on its forward stream
        extra, colorless = None, {}
        side = self._side_stream() if self.use_colorless_loss else None
        if self.use_colorless_loss:
            main = torch.cuda.current_stream()
            if side is not None:
                side.wait_stream(main)
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                # normalize (:317-332) and the rotations (Q, QQ) lead the side branch while the main
                # stream evaluates the gain network; the main solve waits for them only
                if normalize_first:
                    self.normalize(data)
                if fl.M.is_cuda and fl.coupling_matrix_type != CouplingMatrixType.RANDOM:
                    fl.group_rotations()
                if side is not None:
                    ready = torch.cuda.Event()
                    ready.record(side)
                S, _ = net.sub_fdn_group_sums(z)
                # spectral + sparsity (last group only, :305-308), weighted, / world size (position
                # independent terms), values and gradients in two launches
                terms = ColorlessTerms.apply(S, fl.group_rotations(), cfg.use_asym_spectral_loss,
                                             cfg.spectral_loss_weight, cfg.sparsity_loss_weight,
                                             shard_loss_scales(self.world_size, 1, 1.0)['colorless'], True)
                extra = terms[0]
                colorless = {'spectral_loss': terms[1].detach(), 'sparsity_loss': terms[2].detach()}
        filt = self.subband_filter_freq_resp if self.subband_process_config is not None else None
        rgain = net.output_scalars.group_gains(data)
        if side is not None:
            torch.cuda.current_stream().wait_event(ready)
        # the decay losses transform H with irfft(H, n = K) (losses.py:207-213, :442-445), which reads
        # bins 0..(K-1)/2 only: the main branch of the TRAINING step is evaluated on those bins alone
        # (solve, output stage and their backward do half the work; losses and gradients are the
        # same numbers -- the upper bins have exactly zero gradient)
        K = z.shape[-1]
        Ku = (K + 1) // 2 if K % 2 == 1 else K
        zu = z[:Ku]
        Y = net.delay_line_responses(zu)
        rows = data.get('row_index')         # collate(lean="rows"): per-receiver entries are stores
        H = OutputStage.apply(Y, net.output_gains.reshape(-1), rgain.to(torch.float32), n,
                              data['target_early_response'][:, :Ku],
                              None if filt is None else filt[:Ku], rows)
        start, length = self._decay_window(K)
        B = H.shape[0]
        gb = B
        if mask_prenorm is not None:
            maskw, count = mask_prenorm, None
        else:
            maskw, count = (self.criterion[1].draw_mask(length, H.device) if draw_mask
                            else (None, float(length)))
            if maskw is not None and self.world_size > 1:
                dist.broadcast(maskw, src=0, group=self.process_group)   # same mask on all ranks
                count = float(maskw.sum().item())
            if self.world_size > 1:
                nb = torch.tensor([B], device=H.device)
                dist.all_reduce(nb, group=self.process_group)
                gb = int(nb.item())
        edr_t = data.get('edr_target')
        edc_t = data.get('edc_target')
        wf = self.criterion[0].frequency_weights.to(H.device) if cfg.use_frequency_weighting else None
        total, edr_v, edc_v = decay_losses(
            H, data.get('target_rir_response'), win=self.stft_win,
            edr_weight=cfg.edr_loss_weight, edc_weight=cfg.edc_loss_weight,
            edc_start=start, edc_len=length, edc_maskw=maskw, edc_count=count,
            edc_maskw_prenormalised=mask_prenorm is not None, freq_weights=wf,
            reduced_pole_radius=None if self.reduced_pole_radius == 1.0 else self.reduced_pole_radius,
            global_batch=gb,
            edr_target=None if edr_t is None else (edr_t[1], edr_t[2]),
            edc_target=None if edc_t is None else edc_t[1],
            side_stream=self._side_stream2(), unit_grad=True, n_time=K, target_rows=rows)
        losses = {'edc_loss': edc_v, 'edr_loss': edr_v}
        heads = [total]
        if extra is not None:
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
                extra.record_stream(torch.cuda.current_stream())
            heads.append(extra)
            losses.update(colorless)
        # ``defer_total``: the caller back-propagates the heads directly (unit upstream gradients)
        # and forms the reported sum afterwards, off the path to the backward pass
        if defer_total:
            losses['_heads'] = heads
        else:
            losses['_total'] = total if extra is None else total + extra
        return losses

    def _step_losses_module_forward(self, data: Dict, draw_mask: bool, defer_total: bool,
                                    mask_prenorm: Optional[torch.Tensor] = None) -> Dict:
        """Step losses through the model's own forward (SVF output filters: the (B, G, K) filter responses make the
        output stage receiver- AND bin-dependent, model.py:588-592) -- same loss kernels, full bin range.  A
        ``lean="rows"`` batch (dataset-level stores + 'row_index', graph replay) is gathered here for the model;
        the decay targets are still read from the stores through the rows."""
        net, cfg = self.net, self.config
        filt = self.subband_filter_freq_resp if self.subband_process_config is not None else None
        rows = data.get('row_index')
        if 'listener_position' not in data or ('target_rir_response' not in data and 'edr_target' not in data):
            raise ValueError("SVF output filters: the batch needs raw listener positions and decay targets")
        fwd = data
        if rows is not None:
            fwd = dict(data)
            for key in ('listener_position', 'norm_listener_position', 'target_early_response'):
                fwd[key] = data[key].index_select(0, rows)
        out = net(fwd, subband_filter=filt)
        H, H_sub = out if net.use_colorless_loss else (out, None)
        K = H.shape[-1]
        start, length = self._decay_window(K)
        gb = H.shape[0] * self.world_size
        if mask_prenorm is not None:
            maskw, count = mask_prenorm, None
        else:
            maskw, count = (self.criterion[1].draw_mask(length, H.device) if draw_mask else (None, float(length)))
        edr_t, edc_t = data.get('edr_target'), data.get('edc_target')
        total, edr_v, edc_v = decay_losses(
            H, data.get('target_rir_response'), win=self.stft_win, edr_weight=cfg.edr_loss_weight,
            edc_weight=cfg.edc_loss_weight, edc_start=start, edc_len=length, edc_maskw=maskw, edc_count=count,
            edc_maskw_prenormalised=mask_prenorm is not None, global_batch=gb,
            edr_target=None if edr_t is None else (edr_t[1], edr_t[2]),
            edc_target=None if edc_t is None else edc_t[1], n_time=K, target_rows=rows)
        losses = {'edc_loss': edc_v, 'edr_loss': edr_v}
        if self.use_colorless_loss:
            S = H_sub[0].T.contiguous()
            spectral = cfg.spectral_loss_weight * group_spectral_loss(S, cfg.use_asym_spectral_loss)
            sparsity = cfg.sparsity_loss_weight * self.colorless_criterion[1](
                net.feedback_loop.group_rotations()[net.num_groups - 1])
            total = total + (spectral + sparsity) / self.world_size
            losses.update({'spectral_loss': spectral.detach(), 'sparsity_loss': sparsity.detach()})
        if defer_total:
            losses['_heads'] = [total]
        else:
            losses['_total'] = total
        return losses

    def graphed(self, dataset, batch_size: int, mask_source: str = "device",
                mask_seed: Optional[int] = None) -> "GraphedTrainStep":
        """normalize + train_step of a fixed-size batch as one HIP graph replay."""
        return GraphedTrainStep(self, dataset, batch_size, mask_source, mask_seed)

    def train_step(self, data: Dict):
        """normalize is called by the loop, as in the reference (:373-379)."""
        self.optimizer.zero_grad(set_to_none=True)
        losses = self._step_losses(data)
        total = losses.pop('_total')
        losses.pop('_heads', None)
        total.backward()
        if self._allreduce is not None:
            if isinstance(self.optimizer, FlatAdam):
                self.optimizer.pack_grads()
            self._allreduce()
        self.optimizer.step()
        return sum(losses.values()), losses

    @torch.no_grad()
    def valid_step(self, data: Dict):
        losses = self._step_losses(data)
        losses.pop('_total')
        losses.pop('_heads', None)
        return sum(losses.values()), losses

    @torch.no_grad()
    def save_ir(self, input_features: Dict, directory: Optional[str] = None, src_pos: Optional[torch.Tensor] = None,
                rec_pos: Optional[torch.Tensor] = None, filename_prefix: str = "ir", norm: bool = True):
        """Impulse responses of one batch (reference :503-564): h = irfft(H) on the device, the pole-radius
        envelope undone, optional peak normalisation; with ``directory`` one 32-bit float stereo wav (h, h) per
        receiver, named like the reference's files, and the reference's return value (H or (H, H_sub_fdn));
        without it nothing is written and (H, h (B, nfft)) comes back."""
        out = get_response(input_features, self.net)
        h = out[-1]
        if self.reduced_pole_radius is not None and self.reduced_pole_radius != 1.0:
            h = h * torch.pow(torch.tensor(1.0 / self.reduced_pole_radius, device=h.device),
                              torch.arange(h.shape[-1], device=h.device))
        if norm:
            h = h / torch.max(torch.abs(h))
        if directory is None:
            return out[0], h
        rec_pos = input_features['listener_position'] if rec_pos is None else rec_pos
        src_pos = input_features['source_position'] if src_pos is None else src_pos
        hc = h.detach().to(torch.float32).cpu().numpy()
        for r, name in ir_filenames(filename_prefix, src_pos, rec_pos):
            write_wav(os.path.join(str(directory), name), self.net.sample_rate, np.stack((hc[r], hc[r]), axis=1))
        return (out[0], out[1]) if self.use_colorless_loss else out[0]


class DirectionalFDNVarReceiverPosTrainer(Trainer):
    """Directional FDN over a grid of receivers (reference trainer.py:690-921): SH-domain response ->
    sub-band filter -> directional responses (analysis matrix) -> directional EDC loss against the
    common-slope amplitudes, plus the colorless terms."""

    # the sub-FDN branch of the colorless loss on a side stream beside the SH-domain forward (False: one stream)
    concurrent_branches = True
    # directional EDC term on irfft(H_sh) mixed in the time domain (losses.directional_edc_loss.forward_sh) instead of
    # irfft(A H_sh): same numbers to rounding, 3/4 of the transforms at order 2
    mix_in_time = True
    # ... and the SH output stage behind the transform as well (losses.directional_edc_loss.forward_lines): the N line
    # responses are transformed instead of the B (order + 1)^2 receiver responses, which never exist
    lines_in_time = True

    def graphed(self, example_batch: Dict, mask_seed: Optional[int] = None) -> "GraphedModuleStep":
        """train_step on batches shaped like ``example_batch`` as one HIP-graph replay."""
        return GraphedModuleStep(self, example_batch, mask_seed)

    def convert_ambi_rir_to_directional_rir(self, H_sh: torch.Tensor) -> torch.Tensor:
        """einsum('jl,blk->bjk', A_sh, H_sh)  (reference :853-865) as one streaming kernel."""
        return SHToDirectional.apply(self.net.sh_output_scalars.analysis_matrix, H_sh)

    @torch.no_grad()
    def save_ir(self, input_features: Dict, directory: str, src_pos: torch.Tensor, rec_pos: torch.Tensor,
                filename_prefix: str = "ir", norm: bool = True):
        """One multi-channel wav (the SH channels) per receiver (reference :868-921)."""
        out = get_response(input_features, self.net)
        h = out[-1]                                                   # (B, L, nfft)
        if norm:
            h = h / torch.max(torch.abs(h))
        hc = h.detach().to(torch.float32).cpu().numpy()
        for r, name in ir_filenames(filename_prefix, src_pos, rec_pos):
            write_wav(os.path.join(str(directory), name), self.net.sample_rate, hc[r].T)
        return (out[0], out[1]) if self.use_colorless_loss else out[0]

    def _side_stream(self):
        if not self.concurrent_branches or not next(self.net.parameters()).is_cuda:
            return None
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream()
        return self._side

    def _use_lines(self, data: Dict) -> bool:
        net = self.net
        if not (self.mix_in_time and self.lines_in_time and next(net.parameters()).is_cuda):
            return False
        amps = data['target_common_slope_amps']
        J = net.sh_output_scalars.analysis_matrix.shape[0]
        return self.criterion[0].lines_supported(data['z_values'].shape[0], net.num_groups,
                                                 net.num_delay_lines_per_group, J, amps.shape[-1])

    def _forward_two_streams(self, data: Dict, filt, side, lines: bool = False):
        """The module's forward (model.py:1043-1094) with its two branches on two streams: the sub-FDN branch of the
        colorless loss (raw-block solve, group sums, spectral + sparsity terms) shares nothing with the SH-domain branch
        but the parameters and the rotations; its elimination kernels are bound by the LDS crossbar, the other branch's
        transforms by HBM.  autograd replays each backward operator on its forward stream."""
        from .functional import SHOutputStage
        net, cfg = self.net, self.config
        z = data['z_values']
        fl = net.feedback_loop
        fl.new_forward()
        net.batch_size = data['listener_position'].shape[0]
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fl.group_rotations()                         # (Q, QQ: both branches read them)
            ready = torch.cuda.Event()
            ready.record(side)
            H_sub = net.sub_fdn_output(z)
            terms = ColorlessTerms.apply(H_sub[0].T.contiguous(), fl.group_rotations(), cfg.use_asym_spectral_loss,
                                         cfg.spectral_loss_weight, cfg.sparsity_loss_weight,
                                         shard_loss_scales(self.world_size, 1, 1.0)['colorless'], True)
        w = net.sh_output_scalars(data, normalise_weights=True)
        main.wait_event(ready)
        Y = net.delay_line_responses(z, transpose=True)
        if lines:
            return (Y, net.output_gains.reshape(-1), w.to(torch.float32)), terms
        H_sh = SHOutputStage.apply(Y, net.output_gains.reshape(-1), w.to(torch.float32), net.num_groups,
                                   net.num_delay_lines_per_group, filt)
        return H_sh, terms

    def _step_losses(self, data: Dict) -> Dict:
        net, cfg = self.net, self.config
        filt = self.subband_filter_freq_resp if self.subband_process_config is not None else None
        side = self._side_stream() if (self.use_colorless_loss and net.use_colorless_loss) else None
        terms = None
        lines = self._use_lines(data)
        if side is not None:
            H_sh, terms = self._forward_two_streams(data, filt, side, lines)
            H_sub = None
        elif lines:
            H_sh = net.output_stage_inputs(data)
            H_sub = net.sub_fdn_output(data['z_values']) if net.use_colorless_loss else None
        else:
            out = net(data, subband_filter=filt)
            H_sh, H_sub = out if net.use_colorless_loss else (out, None)
        # (the weight rides the kernel's gradient scale and the term enters the total with factor 1: the backward skips
        # the pass that would multiply the gradient by the upstream scalar)
        crit = self.criterion[0]
        if lines:
            # the whole chain between the line responses and the EDC functional is linear: N transforms, not B (order + 1)^2
            Y, c, w = H_sh
            edc = crit.forward_lines(Y, c, w, net.num_groups, net.num_delay_lines_per_group, filt,
                                     net.sh_output_scalars.analysis_matrix, data['target_common_slope_amps'],
                                     weight=cfg.edc_loss_weight, unit_grad=True)
        elif self.mix_in_time:
            # SH -> directional conversion behind the inverse transform (linear maps commute): C transforms per receiver
            edc = crit.forward_sh(H_sh, net.sh_output_scalars.analysis_matrix, data['target_common_slope_amps'],
                                  weight=cfg.edc_loss_weight, unit_grad=True)
        else:
            H_dir = self.convert_ambi_rir_to_directional_rir(H_sh)
            edc = crit(H_dir, data['target_common_slope_amps'], weight=cfg.edc_loss_weight, unit_grad=True)
        losses = {'edc_loss': edc.detach()}
        total = edc
        if self.use_colorless_loss:
            # spectral + sparsity (last group only, reference :298-313), weighted, / world size: values and both
            # gradients in two launches (the tensor-operator form took 11 forward and 8 backward)
            if terms is None:
                terms = ColorlessTerms.apply(H_sub[0].T.contiguous(), net.feedback_loop.group_rotations(),
                                             cfg.use_asym_spectral_loss, cfg.spectral_loss_weight,
                                             cfg.sparsity_loss_weight,
                                             shard_loss_scales(self.world_size, 1, 1.0)['colorless'], True)
            else:
                torch.cuda.current_stream().wait_stream(side)
                for t_ in terms:
                    t_.record_stream(torch.cuda.current_stream())
            total = total + terms[0]
            losses.update({'spectral_loss': terms[1].detach(), 'sparsity_loss': terms[2].detach()})
        losses['_total'] = total
        return losses

    def train_step(self, data: Dict):
        self.optimizer.zero_grad(set_to_none=True)
        losses = self._step_losses(data)
        total = losses.pop('_total')
        total.backward()
        if self._allreduce is not None:
            if isinstance(self.optimizer, FlatAdam):
                self.optimizer.pack_grads()
            self._allreduce()
        self.optimizer.step()
        # (one process: the reported sum of the terms IS the total that was back-propagated)
        return (total.detach() if self.world_size == 1 else sum(losses.values())), losses

    @torch.no_grad()
    def valid_step(self, data: Dict):
        losses = self._step_losses(data)
        total = losses.pop('_total')
        return (total if self.world_size == 1 else sum(losses.values())), losses


class SinglePosTrainer(Trainer):
    """One source-receiver pair (reference trainer.py:570-684; the reference's own train_step has
    unbound variables on the colorless / no-sub-band branches, SURVEY §8c -- this is the working
    counterpart).  ``data``: dict of (K,) tensors z_values, target_rir_response, target_early_response."""

    @torch.no_grad()
    def save_ir(self, data: Dict, directory: str, filename_prefix: str = 'ir', norm: bool = False):
        """The modelled impulse response as one stereo wav (reference :664-684)."""
        h = get_response(data, self.net)[-1].reshape(-1)
        if self.reduced_pole_radius is not None and self.reduced_pole_radius != 1.0:
            h = h * torch.pow(torch.tensor(1.0 / self.reduced_pole_radius, device=h.device),
                              torch.arange(h.shape[-1], device=h.device))
        if norm:
            h = h / torch.max(torch.abs(h))
        hc = h.detach().to(torch.float32).cpu().numpy()
        write_wav(os.path.join(str(directory), filename_prefix + '.wav'), self.net.sample_rate,
                  np.stack((hc, hc), axis=1))
        return h

    def _step_losses(self, data: Dict) -> Dict:
        net, cfg = self.net, self.config
        out = net(data)
        H, H_sub = out if net.use_colorless_loss else (out, None)
        if self.subband_process_config is not None:
            H = H * self.subband_filter_freq_resp
        K = H.shape[-1]
        start, length = self.criterion[1].window(K)
        maskw, count = self.criterion[1].draw_mask(length, H.device)
        total, edr_v, edc_v = decay_losses(
            H, data['target_rir_response'], win=self.stft_win, edr_weight=cfg.edr_loss_weight,
            edc_weight=cfg.edc_loss_weight, edc_start=start, edc_len=length, edc_maskw=maskw,
            edc_count=count,
            reduced_pole_radius=None if self.reduced_pole_radius == 1.0 else self.reduced_pole_radius)
        losses = {'edc_loss': edc_v, 'edr_loss': edr_v}
        if self.use_colorless_loss:
            S = H_sub[0].T.contiguous()
            spectral = cfg.spectral_loss_weight * group_spectral_loss(S, cfg.use_asym_spectral_loss)
            fl = net.feedback_loop
            sparsity = cfg.sparsity_loss_weight * self.colorless_criterion[1](
                fl.group_rotations()[net.num_groups - 1])
            total = total + spectral + sparsity
            losses.update({'spectral_loss': spectral.detach(), 'sparsity_loss': sparsity.detach()})
        losses['_total'] = total
        return losses

    def train_step(self, data: Dict):
        self.optimizer.zero_grad(set_to_none=True)
        losses = self._step_losses(data)
        total = losses.pop('_total')
        total.backward()
        self.optimizer.step()
        return sum(losses.values()), losses

    @torch.no_grad()
    def valid_step(self, data: Dict):
        losses = self._step_losses(data)
        losses.pop('_total')
        return sum(losses.values()), losses


class GraphedModuleStep:
    """``trainer.train_step(batch)`` of a trainer whose forward goes through the model's own module (directional
    and single-position models) captured into ONE HIP graph.  The batch lives in static device buffers that
    ``__call__`` refreshes in place; warm-up and capture run on ONE private stream -- autograd pins every
    parameter's AccumulateGrad node to the stream of its first backward, and a capture that has to hop to another
    stream and back dies in hipStreamEndCapture (ROCm 7.2).  Needs ``capturable=True`` (flat Adam with device-side
    step counter and learning rates) and a loss path without host reads: with ``use_edc_mask`` the directional trainer's
    time mask (reference losses.py:287-292, :355-360: drawn on the host every step) is drawn INSIDE the graph by the
    counter-based device generator (``mask_seed``; the bits of ``gfdn_draw_mask`` at (seed, replay number))."""

    def __init__(self, trainer, example_batch: Dict, mask_seed: Optional[int] = None):
        if not trainer.capturable:
            raise ValueError("build the trainer with capturable=True to replay steps from a graph")
        self.mask_state = None
        if getattr(trainer.config, 'use_edc_mask', False):
            crit = trainer.criterion[0]
            if not isinstance(crit, directional_edc_loss):
                raise NotImplementedError("GraphedModuleStep: a device-side EDC mask exists for the directional loss only")
            K = example_batch['z_values'].shape[-1]
            L = min(crit.edc_len_samps, 2 * (K - 1) - crit.mixing_time_samps)
            dev = example_batch['z_values'].device
            if mask_seed is None:
                mask_seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.long).item())   # torch.manual_seed governs it
            self.mask_seed = int(mask_seed)
            self.mask_state = torch.zeros(1, dtype=torch.long, device=dev)
            self.maskw = torch.zeros(L, dtype=torch.float32, device=dev)
            self._crit = crit              # (the device generator is installed on the criterion around warm-up and capture
            #                                 only -- device_mask_scope --: eager steps and validation on the same trainer
            #                                 keep the host draw that torch.manual_seed governs)
        self.tr = trainer
        self.batch = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        self.stream = None                 # (created by capture(): torch hands streams out of a pool of 32 round-robin, and
        #                                     a DirectionalBank's steps never capture on their own)
        self.graph = None
        self.out = None

    @contextlib.contextmanager
    def device_mask_scope(self):
        """The criterion draws its EDC time mask from this step's device generator inside the block"""
        if self.mask_state is None:
            yield
            return
        prev = self._crit.device_mask
        self._crit.device_mask = (self.mask_seed, self.mask_state, self.maskw)
        try:
            yield
        finally:
            self._crit.device_mask = prev

    def capture(self):
        with self.device_mask_scope():
            return self._capture()

    def _capture(self):
        tr = self.tr
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        params = list(tr.net.parameters())
        saved_p = [p.detach().clone() for p in params]
        saved_s = [t.detach().clone() for t in tr.optimizer.state_tensors()]
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(3):
                tr.train_step(self.batch)
            for p, sp in zip(params, saved_p):                 # the warm-up leaves no trace
                p.data.copy_(sp)
            for t, st in zip(tr.optimizer.state_tensors(), saved_s):
                t.copy_(st)
            if self.mask_state is not None:
                self.mask_state.zero_()                        # (the warm-up draws leave no trace either)
            tr.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            total, parts = tr.train_step(self.batch)
        self.out = (total, parts)
        from .functional import FrequencyGrid
        self._grids = list(FrequencyGrid._cache.values())      # (raw pointers into the grids were recorded)
        torch.cuda.synchronize()
        return self

    def __del__(self):
        try:                                 # (as GraphedTrainStep.__del__)
            if self.graph is not None:
                torch.cuda.synchronize()
        except Exception:        # noqa: BLE001
            pass

    def __call__(self, batch: Optional[Dict] = None):
        if batch is not None:
            for k, v in batch.items():
                dst = self.batch.get(k)
                if torch.is_tensor(v) and torch.is_tensor(dst) and v.data_ptr() != dst.data_ptr():
                    dst.copy_(v, non_blocking=True)
        if self.graph is None:
            self.capture()
        self.graph.replay()
        return self.out


class DirectionalBank:
    """The octave bands' directional trainers (reference run_subband_training_treble.py:175-204 builds one
    ``DirectionalFDNVarReceiverPosTrainer`` per band and trains them one after another) stepped TOGETHER: every band's
    ``train_step`` is captured into ONE HIP graph, the bands dealt round-robin onto ``lanes`` streams inside the capture, so
    that one replay steps all bands and one band's latency-bound stretches (the crossbar-bound 9 x 9 eliminations, a dozen
    launches of a few microseconds) run beside another band's bandwidth-bound kernels without the host launching seven
    graphs.  The bands stay independent models with their own parameters, optimiser state and loss modules -- this is a bank
    at the level of the GRAPH, not of the launch (a launch per stage for all bands would need the band dimension in every
    kernel of the directional step, as ``bandbank.BandBank`` has for the omnidirectional model).
    ``batches``: one batch dict per band; they live in static device buffers that ``__call__`` refreshes in place."""

    def __init__(self, trainers, example_batches, lanes: int = 2, mask_seed: Optional[int] = None):
        if len(trainers) != len(example_batches) or not trainers:
            raise ValueError("DirectionalBank: one example batch per band trainer")
        self.steps = [GraphedModuleStep(tr, b, None if mask_seed is None else mask_seed + q)
                      for q, (tr, b) in enumerate(zip(trainers, example_batches))]
        # torch hands streams out of a pool of 32 per device, round-robin: in a process that has created many, two "new"
        # streams can be the SAME stream -- a lane that aliases another lane or a band's side stream turns the fork / join
        # pattern below into one the capture rejects ("unjoined work").  Every stream of the bank is checked to be distinct.
        taken = set()

        def fresh():
            for _ in range(64):
                st = torch.cuda.Stream()
                if st.cuda_stream not in taken:
                    taken.add(st.cuda_stream)
                    return st
            raise RuntimeError("DirectionalBank: no distinct stream left in the pool")

        self.lanes = [fresh() for _ in range(max(1, min(int(lanes), len(trainers))))]
        # (the bank rearranges its trainers' streams and branches: what they had is put back by close())
        self._restore = [(tr, getattr(tr, 'concurrent_branches', False), getattr(tr, '_side', None)) for tr in trainers]
        for q, tr in enumerate(trainers):
            if q % len(self.lanes) == 0 and getattr(tr, 'concurrent_branches', False):
                tr._side = fresh()                             # (the first lane's bands keep their side stream)
        self.root = self.lanes[0]          # (the capture's origin is the first lane: a separate origin stream that only
        #                                     forks and joins makes hipStreamEndCapture of ROCm 7.2 segfault)
        # ... and so does a forked lane that forks again: the bands on the lanes beside the first run their colorless branch
        # on the lane itself, not on the trainer's side stream (the bands of the OTHER lane run beside it anyway)
        for q, tr in enumerate(trainers):
            if q % len(self.lanes):
                tr.concurrent_branches = False
        self.graph = None
        self.out = None

    def close(self):
        """Hands the trainers back as they came: their own side streams and ``concurrent_branches`` (the captured graph keeps
        replaying on the bank's lanes; eager steps of a trainer run as before the bank was built)."""
        for tr, branches, side in self._restore:
            if hasattr(tr, 'concurrent_branches'):
                tr.concurrent_branches = branches
            if side is not None or hasattr(tr, '_side'):
                tr._side = side
        self._restore = []

    @property
    def batches(self):
        return [st.batch for st in self.steps]

    def _lane(self, q: int):
        return self.lanes[q % len(self.lanes)]

    def capture(self):
        with contextlib.ExitStack() as scopes:
            for st in self.steps:
                scopes.enter_context(st.device_mask_scope())
            return self._capture()

    def _capture(self):
        cur = torch.cuda.current_stream()
        # warm-up per band ON ITS LANE (autograd pins every parameter's AccumulateGrad node to the stream of its first
        # backward), leaving no trace -- as GraphedModuleStep.capture
        for q, st in enumerate(self.steps):
            tr, lane = st.tr, self._lane(q)
            params = list(tr.net.parameters())
            saved_p = [p.detach().clone() for p in params]
            saved_s = [t.detach().clone() for t in tr.optimizer.state_tensors()]
            lane.wait_stream(cur)
            with torch.cuda.stream(lane):
                for _ in range(3):
                    tr.train_step(st.batch)
                for p, sp in zip(params, saved_p):
                    p.data.copy_(sp)
                for t, s_ in zip(tr.optimizer.state_tensors(), saved_s):
                    t.copy_(s_)
                if st.mask_state is not None:
                    st.mask_state.zero_()
                tr.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        outs = [None] * len(self.steps)
        with torch.cuda.graph(self.graph, stream=self.root):
            for lane in self.lanes[1:]:
                lane.wait_stream(self.root)                    # fork
            for q, st in enumerate(self.steps):
                with torch.cuda.stream(self._lane(q)):
                    outs[q] = st.tr.train_step(st.batch)
            for lane in self.lanes[1:]:
                self.root.wait_stream(lane)                    # join
        self.out = outs
        from .functional import FrequencyGrid
        self._grids = list(FrequencyGrid._cache.values())
        torch.cuda.synchronize()
        return self

    def __del__(self):
        try:
            if self.graph is not None:
                torch.cuda.synchronize()
        except Exception:        # noqa: BLE001
            pass

    def __call__(self, batches=None):
        if batches is not None:
            for st, batch in zip(self.steps, batches):
                if batch is None:
                    continue
                for k, v in batch.items():
                    dst = st.batch.get(k)
                    if torch.is_tensor(v) and torch.is_tensor(dst) and v.data_ptr() != dst.data_ptr():
                        dst.copy_(v, non_blocking=True)
        if self.graph is None:
            self.capture()
        self.graph.replay()
        return self.out


def _graph_node_count(graph: "torch.cuda.CUDAGraph") -> Optional[int]:
    """Number of nodes of a captured graph (hipGraphGetNodes on the raw handle; the graph must have been created with
    keep_graph=True).  None if the runtime does not answer."""
    import ctypes
    try:
        hip = ctypes.CDLL('libamdhip64.so')
        n = ctypes.c_size_t(0)
        rc = hip.hipGraphGetNodes(ctypes.c_void_p(graph.raw_cuda_graph()), None, ctypes.byref(n))
        return int(n.value) if rc == 0 else None
    except (OSError, AttributeError, RuntimeError):
        return None


class GraphedTrainStep:
    """One optimiser step (collate -> normalize -> forward -> losses -> backward -> [all-reduce] ->
    Adam) captured into HIP graphs and replayed: ~400 kernel launches per step collapse into one
    (two with the RCCL all-reduce of the flat gradient buffer between them) graph launch, which removes the host launch
    overhead that dominates the eager step (profiles/).

    Per step the host only writes one static device buffer, the receiver indices of the batch.
    The EDC time mask (losses.py:221-223) is drawn INSIDE the graph by a counter-based device
    generator (``mask_source='device'``, the default): the CPU draw costs ~0.5 ms per step on the
    host, which is as long as the whole GPU step, and in a data-parallel job every rank derives
    the same mask from the shared seed with no broadcast.  ``mask_source='host'`` keeps the
    reference's CPU draw (uniform_ then bernoulli from torch's global generator), staged through
    pinned memory.  Requires a trainer built with ``capturable=True`` and a fixed batch size."""

    def __init__(self, trainer: "VarReceiverPosTrainer", dataset, batch_size: int,
                 mask_source: str = "device", mask_seed: Optional[int] = None):
        if not trainer.capturable:
            raise ValueError("build the trainer with capturable=True to replay steps from a graph")
        if mask_source not in ("device", "host"):
            raise ValueError("mask_source must be 'device' or 'host'")
        self.tr, self.ds, self.B = trainer, dataset, batch_size
        self.mask_source = mask_source
        dev = dataset.device
        K = dataset.rir_mag_response.shape[-1]
        self.start, self.length = trainer._decay_window(K)
        # a band bank (bandbank.BandBankTrainer) steps ``num_bands`` independent models at once: the
        # batch holds batch_size / num_bands receivers per model, and loss heads are (bands,) vectors
        self.num_bands = getattr(trainer, 'num_bands', 1)
        if batch_size % self.num_bands:
            raise ValueError("batch size must be a multiple of the number of bands")
        self.gb = batch_size // self.num_bands * trainer.world_size
        # replayed launches read the targets through the static index buffer: they must come from
        # the dataset-level store (a by-pointer cache would go stale under replay).  A band bank whose bands have
        # different EDC windows keeps targets of per-band lengths and one row of mask weights per band
        tw = trainer._target_window(K) if hasattr(trainer, '_target_window') else (self.start, self.length)
        dataset.precompute_decay_targets(trainer.stft_win, *tw)         # (no-op when the stores hold these targets)
        self.band_len = None
        if hasattr(trainer, '_item_windows'):
            _, self.band_len = trainer._item_windows(K, batch_size // self.num_bands, dev)
        self.idx = torch.zeros(batch_size, dtype=torch.long, device=dev)
        if self.band_len is None:
            self.maskw = torch.full((self.length,), 1.0 / (self.gb * self.length), dtype=torch.float32, device=dev)
        else:
            self.maskw = trainer._band_mask_rows(None, K, self.gb, dev)          # (bands, length): no mask = all kept
        self._K = K
        # ring of pinned staging buffers: the async H2D copy of step i is only executed when the
        # stream reaches it, so its source must not be rewritten by the host preparing step i+1
        self._ring = [(torch.empty(self.length, dtype=torch.float32).pin_memory(),
                       torch.empty(batch_size, dtype=torch.long).pin_memory(), torch.cuda.Event())
                      for _ in range(4)]
        self._ring_pos = 0
        self.mask_state = torch.zeros(1, dtype=torch.long, device=dev)     # draws made so far
        self._one = torch.ones(() if self.num_bands == 1 else (self.num_bands,), dtype=torch.float32,
                               device=dev)                                  # root gradient of the heads
        if mask_seed is None:
            seed_t = torch.randint(0, 2 ** 62, (1,), dtype=torch.long)      # torch.manual_seed governs it
            if trainer.world_size > 1:                                      # one seed for all ranks, once
                seed_t = seed_t.to(dev)
                dist.broadcast(seed_t, src=0, group=trainer.process_group)
            mask_seed = int(seed_t.item())
        self.mask_seed = int(mask_seed)
        self.graph_a = self.graph_b = self.graph_p = None
        self._tail_in_graph = False
        self.collective_probe_nodes = None        # nodes the probe's captured all-reduce left in its graph (None: no probe)
        self.losses = None
        # receiver schedule (load_schedule / run_next): the batches of an epoch live on the device and every step ends
        # by fetching the next step's receivers into ``idx`` -- no host copy in front of a replay
        self.sched_cap = 1024
        self.sched = torch.zeros((self.sched_cap, batch_size), dtype=torch.long, device=dev)
        self.sched_state = torch.tensor([0, 1], dtype=torch.long, device=dev)        # {position, length}

    def _ensure_records(self):
        """The explicit bank step's graph starts from the records its previous replay left (bankstep.FusedBankStep): if
        anything else has touched the parameters since, rebuild them in front of the replay."""
        fused = getattr(self.tr, '_fused', None)
        if fused is None:
            return
        if self._tail_in_graph:
            fused.ensure_records()
        else:
            fused.invalidate_records()        # (this graph evaluates its own records and leaves the parameters changed)

    # -- receiver schedule --------------------------------------------------------------------
    def _pick_next(self):
        ops.pick_rows(self.sched, self.sched_state, self.idx)

    def load_schedule(self, batches):
        """Upload the receivers of the next ``len(batches)`` steps (each ``batch_size`` dataset rows; at most
        ``sched_cap`` steps) and point the step at the first of them; ``run_next()`` then replays one step per call.
        The reference's DataLoader likewise fixes an epoch's batches when the epoch starts (trainer.py:373-379)."""
        # (a (steps, batch) int64 tensor -- pinned host memory for an asynchronous upload -- is taken as it is; lists of indices
        # are converted here, which costs ~0.2 ms for 20 x 224 Python integers.  A PINNED source is read by the copy engine
        # AFTER this call returns: the caller must leave it unmodified until the current stream has passed this point -- an
        # event is kept in ``sched_uploaded`` for callers that refill one pinned table)
        t = batches.to(torch.long) if torch.is_tensor(batches) else torch.as_tensor([list(b) for b in batches], dtype=torch.long)
        n = t.shape[0]
        if n == 0 or n > self.sched_cap or t.shape[1] != self.B:
            raise ValueError(f"load_schedule: 1..{self.sched_cap} batches of {self.B} receivers")
        self.sched[:n].copy_(t, non_blocking=True)
        self.sched_uploaded = torch.cuda.Event()
        self.sched_uploaded.record()
        self.sched_state.copy_(torch.tensor([0, n], dtype=torch.long))
        self._pick_next()                                 # idx <- first batch, position 1
        return n

    def run_next(self):
        """One optimiser step on the next batch of the loaded schedule (wraps around at its end); returns the static
        loss tensors (valid until the next step)."""
        if self.graph_a is None:
            self.capture(None)
        self._load_inputs(None)
        self._ensure_records()
        self.graph_a.replay()
        if self.graph_b is not None:
            self.tr._allreduce()
            self.graph_b.replay()
        return self.losses

    def run_schedule(self, batches):
        """Generator over the steps of ``batches`` (any number: uploaded in chunks of ``sched_cap``).  Where the
        pipelined chain applies (:meth:`_pipe_ok`) the steps run ``pipe_steps`` at a time from one graph and the rest
        from the single-step graph -- same numbers either way."""
        if not torch.is_tensor(batches):
            batches = list(batches)
        for i0 in range(0, len(batches), self.sched_cap):
            n = self.load_schedule(batches[i0:i0 + self.sched_cap])
            done = 0
            S = self.pipe_steps
            if self._pipe_ok() and n >= S:
                if self.graph_p is None:
                    self.capture_pipe()
                self._pipe_prologue()
                while n - done >= S:
                    self.graph_p.replay()
                    self.tr._fused.invalidate_records()      # (the chain evaluates its own records: the kept ones are stale)
                    for losses in self.pipe_losses:
                        yield losses
                    done += S
            for _ in range(n - done):
                yield self.run_next()

    # -- pipelined chain of explicit bank steps (bankstep.StepPipe) -----------------------------
    # steps per graph of the pipelined chain (even; 0: single-step graphs only).  OFF by default: measured on the 7-band
    # step, same box, 400 steps: 0.681 ms single-step graphs, 0.692 / 0.688 / 0.688 ms with 2 / 4 / 8 steps per graph.
    # Under rocprofv3 the interior steps of a chain are 35 us shorter (no wait for the gains at the head, Adam straight
    # behind the last gradient kernel, no join), but the chain's last step waits for the side stream's run-ahead work
    # (+60 us per graph) and without the profiler's per-kernel serialisation the interior gain is not there either.
    pipe_steps = 0

    def _pipe_ok(self) -> bool:
        tr = self.tr
        return (self.pipe_steps >= 2 and self.pipe_steps % 2 == 0 and getattr(tr, '_fused', None) is not None
                and tr._allreduce is None and tr._stream('_side2') is not None
                and not (tr.criterion[1].use_mask and self.mask_source == "host"))

    def _mlp_args(self):
        bank = self.tr.net
        Hh, n_hidden, _, lo, hi = bank._mlp_cfg
        return (self.ds.norm_listener_position, bank._freq_pi, bank.output_scalars_w.detach(), Hh, n_hidden,
                bank.num_groups, lo, hi, self.idx, self.num_bands)

    def _pipe_prologue(self):
        """Receiver gains of the batch in ``idx`` at the current parameters -> the chain's first buffer set."""
        ops.mlp_gains_fwd(*self._mlp_args(), out=self._pipe_bufs[0])

    def capture_pipe(self):
        """Record ``pipe_steps`` explicit steps as ONE graph with the side stream running ahead (StepPipe)."""
        from .bankstep import StepPipe
        if self.graph_a is None:
            self.capture(None)                      # (warm-up of every kernel, lazy initialisations)
        bank = self.tr.net
        Hh, n_hidden = bank._mlp_cfg[0], bank._mlp_cfg[1]
        if bank.mixed_networks:
            raise NotImplementedError("a pipelined chain of steps takes a bank of equal gain networks")
        dev, G, nl = self.idx.device, bank.num_groups, 1 + n_hidden
        mk = lambda: (torch.empty((self.B, G), dtype=torch.float32, device=dev),
                      torch.empty((self.B, nl, Hh), dtype=torch.float32, device=dev),
                      torch.empty((self.B, nl), dtype=torch.float32, device=dev))
        self._pipe_bufs = [mk(), mk()]
        pipe = StepPipe(self._pipe_bufs, self.pipe_steps)
        rng_state = torch.get_rng_state()
        torch.cuda.synchronize()
        self.graph_p = torch.cuda.CUDAGraph()
        losses = []
        with torch.cuda.graph(self.graph_p, stream=self._capture_stream):
            for _ in range(self.pipe_steps):
                losses.append(dict(self._fused_fwd_bwd(opt_step=True, pipe=pipe)))
        self.pipe_losses = losses
        from .functional import FrequencyGrid
        self._grids = list(FrequencyGrid._cache.values())
        torch.set_rng_state(rng_state)
        return self

    # -- pieces -------------------------------------------------------------------------------
    def _fwd_bwd(self):
        tr = self.tr
        if self.mask_source == "device" and tr.criterion[1].use_mask:
            # (measured: drawing it on the EDC side stream adds a third branch at the head of the graph
            # and the executor then serialises the whole front: +0.05 ms)
            ops.draw_mask(self.mask_seed, self.mask_state, self.length, 1.0 / self.gb, out=self.maskw,
                          band_len=self.band_len)
        # no gather: the kernels read the dataset-level stores through the static index buffer
        batch = self.ds.collate(self.idx, lean="rows")
        tr.optimizer.zero_grad(set_to_none=True)
        losses = tr._step_losses(batch, mask_prenorm=self.maskw, normalize_first=True, defer_total=True)
        heads = losses.pop('_heads')
        torch.autograd.backward(heads, [self._one] * len(heads))      # no ones-fill, no add in front
        with torch.no_grad():
            losses['_total'] = heads[0].detach() if len(heads) == 1 else heads[0].detach() + heads[1].detach()
        return losses

    def _fused_fwd_bwd(self, opt_step: bool, pipe=None):
        """The band bank's explicit launch sequence (bankstep.FusedBankStep): mask draw, normalize, forward, losses,
        backward into the flat gradient buffer [, all-reduce, Adam]."""
        tr = self.tr
        draw = None
        if self.mask_source == "device" and tr.criterion[1].use_mask:
            draw = lambda: ops.draw_mask(self.mask_seed, self.mask_state, self.length, 1.0 / self.gb, out=self.maskw,
                                         band_len=self.band_len)
        batch = self.ds.collate(self.idx, lean="rows")
        return tr._fused.run(batch, self.maskw, 1.0, normalize_first=True, train=True, allreduce=tr._allreduce,
                             opt_step=opt_step, mask_draw=draw, tail=self._pick_next, pipe=pipe)

    def _eager(self):
        if getattr(self.tr, '_fused', None) is not None:
            return self._fused_fwd_bwd(opt_step=True)
        losses = self._fwd_bwd()
        self.tr.optimizer.pack_grads()
        if self.tr._allreduce is not None:
            self.tr._allreduce()
        self.tr.optimizer.step()
        self._pick_next()
        return losses

    # -- capture ------------------------------------------------------------------------------
    def capture(self, indices):
        """Warm up (eagerly, on a side stream), restore every parameter / optimizer tensor IN PLACE
        so that the warm-up leaves no trace, then record the graphs."""
        tr = self.tr
        rng_state = torch.get_rng_state()                  # capture is RNG-neutral
        self._load_inputs(indices)
        saved_sched = (self.sched_state.clone(), self.idx.clone())     # (the warm-up steps advance the schedule)
        params = [p for p in tr.net.parameters()]
        saved_p = [p.detach().clone() for p in params]
        state_tensors = tr.optimizer.state_tensors
        saved_s = [t.detach().clone() for t in state_tensors()]   # empty for a fresh optimizer
        # warm-up AND capture on ONE private stream: autograd pins every parameter's AccumulateGrad node to the stream of
        # its first backward; a capture on another stream then hops to that stream and back for every parameter
        side = self._capture_stream = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self._eager()
            # undo the warm-up IN PLACE (the graphs will reference these very tensors)
            for p, sp in zip(params, saved_p):
                p.data.copy_(sp)
            if saved_s:
                for t, st in zip(state_tensors(), saved_s):
                    t.copy_(st)
            else:
                for t in state_tensors():
                    t.zero_()                              # fresh Adam state: zeros, step 0
            self.mask_state.zero_()
            self.sched_state.copy_(saved_sched[0])
            self.idx.copy_(saved_sched[1])
            tr.optimizer.zero_grad(set_to_none=True)
            if getattr(tr, '_fused', None) is not None:
                # the explicit step keeps the records of the current parameters between steps (its fused tail leaves the
                # next step's): those of the restored parameters, so that the capture holds no records launch
                tr._fused.prime_records()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        fused = getattr(tr, '_fused', None) is not None
        # One step structure for the WHOLE group: every rank probes for itself (the probe never raises), then the
        # verdicts are reduced with MIN -- a rank that cannot capture the collective takes every other rank with it to
        # the two-graph step, so no two ranks ever pair a captured all-reduce with an eager one (a hang)
        in_graph = False
        if tr._allreduce is not None and getattr(tr, 'allreduce_in_graph', False):
            in_graph = self._all_ranks_agree(self._collective_is_capturable())
        self.allreduce_in_graph = in_graph
        self.graph_a = torch.cuda.CUDAGraph()
        if tr._allreduce is None or in_graph:
            # ONE graph: forward, backward, [all-reduce of the gradient bucket (RCCL is capturable),] Adam
            with torch.cuda.graph(self.graph_a, stream=side):
                if fused:
                    self.losses = self._fused_fwd_bwd(opt_step=True)
                else:
                    self.losses = self._fwd_bwd()
                    tr.optimizer.pack_grads()
                    if tr._allreduce is not None:
                        tr._allreduce()
                    tr.optimizer.step()
                    self._pick_next()
        else:
            # forward + backward + pack | all-reduce of the gradient bucket (eager RCCL) | Adam
            with torch.cuda.graph(self.graph_a, stream=side):
                if fused:
                    self.losses = self._fused_fwd_bwd(opt_step=False)
                else:
                    self.losses = self._fwd_bwd()
                    tr.optimizer.pack_grads()
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), stream=side):
                if fused:
                    red = tr._fused.finish(lambda: None)          # (the collective itself runs between the replays)
                    self._apply_reduced(red)
                else:
                    tr.optimizer.step()
                    self._pick_next()
        self.losses = {k: v for k, v in self.losses.items()}
        # whether the captured step ends by leaving the next step's records (bankstep.FusedBankStep: fused tail) -- such a
        # graph holds no records launch and relies on _ensure_records() in front of every replay
        self._tail_in_graph = bool(fused and tr._fused._rec_valid)
        # the recorded launches hold raw pointers into the frequency grids (turns / log radius): keep
        # the grid objects alive as long as the graphs (the by-pointer cache may drop them)
        from .functional import FrequencyGrid
        self._grids = list(FrequencyGrid._cache.values())
        torch.set_rng_state(rng_state)
        return self

    def __del__(self):
        # a graph executable must not be destroyed while its last launch is still in flight (its kernel-argument
        # buffers go with it): drain the device first
        try:
            if self.graph_a is not None or self.graph_p is not None:
                torch.cuda.synchronize()
        except Exception:        # noqa: BLE001 -- interpreter shutdown
            pass

    def _apply_reduced(self, red):
        """Split-graph data-parallel step: the second graph rebuilt the decay losses from the reduced slots."""
        if red is None:
            return
        sums, total = red
        nb = self.num_bands
        self.losses = dict(self.losses)
        self.losses['edr_loss'] = sums[:, 1] if nb > 1 else sums[1]
        self.losses['edc_loss'] = sums[:, 2] if nb > 1 else sums[2]
        self.losses['_total'] = total

    def _all_ranks_agree(self, ok: bool) -> bool:
        """MIN of the ranks' verdicts (one small collective, issued by every rank whatever its own verdict)."""
        tr = self.tr
        pg = getattr(tr, 'process_group', None)
        if not dist.is_initialized() or dist.get_world_size(pg) == 1:
            return bool(ok)
        on_host = dist.get_backend(pg) != 'nccl'
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device='cpu' if on_host else self.idx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=pg)
        agreed = bool(int(t.item()))
        if ok and not agreed and getattr(tr, 'rank', 0) == 0:
            print("[diffgfdn_amd] another rank cannot capture the all-reduce: two-graph step on every rank")
        return agreed

    def _collective_is_capturable(self) -> bool:
        """Probe: capture the trainer's collective on a scratch tensor in a throw-away graph and replay it.  Any
        error -> the step falls back to two graphs with the collective between them (the same result)."""
        tr = self.tr
        pg = getattr(tr, 'process_group', None)
        if not dist.is_initialized() or dist.get_backend(pg) != 'nccl':
            return False               # (gloo rehearsals: host-side collectives cannot be captured)
        try:
            probe = torch.ones(64, dtype=torch.float32, device=self.idx.device)
            dist.all_reduce(probe, group=pg)                     # communicator set up outside the capture
            torch.cuda.synchronize()
            probe.fill_(1.0)
            g = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(g):
                dist.all_reduce(probe, group=pg)
            # how many nodes the captured collective left in the graph (0 on a one-rank group, where the library
            # makes an in-place all-reduce a no-op: such a probe says nothing about capturing a collective)
            self.collective_probe_nodes = _graph_node_count(g)
            g.replay()
            torch.cuda.synchronize()
            ok = bool((probe == float(dist.get_world_size(pg))).all().item())
            del g
            return ok
        except Exception as e:                                   # noqa: BLE001 -- any failure means "do not capture"
            if getattr(tr, 'rank', 0) == 0:
                print(f"[diffgfdn_amd] all-reduce not capturable here ({type(e).__name__}: {e}); two-graph step")
            torch.cuda.synchronize()
            return False

    def _load_inputs(self, indices):
        tr = self.tr
        host_mask, host_idx, ev = self._ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(self._ring)
        crit = tr.criterion[1]
        if indices is None and not (crit.use_mask and self.mask_source == "host"):
            return                           # schedule mode: the previous step (or load_schedule) has filled idx
        ev.synchronize()                     # the copies that last used this slot have executed
        if indices is not None:
            host_idx.copy_(torch.as_tensor(list(indices), dtype=torch.long))
            self.idx.copy_(host_idx, non_blocking=True)
        if crit.use_mask and self.mask_source == "host":
            keep = torch.bernoulli(torch.empty(self.length).uniform_(0, 1))
            if self.band_len is not None:
                # bands with different windows: one draw of the longest, truncated per band (as the device draw)
                if tr.world_size > 1:
                    keep = keep.to(self.maskw.device)
                    dist.broadcast(keep, src=0, group=tr.process_group)
                self.maskw.copy_(tr._band_mask_rows(keep, self._K, self.gb, self.maskw.device))
            elif tr.world_size > 1:
                # one mask for all ranks: rank 0's draw wins
                keep = keep.to(self.maskw.device)
                dist.broadcast(keep, src=0, group=tr.process_group)
                cnt = keep.sum()
                self.maskw.copy_(keep / (cnt * self.gb))
            else:
                torch.div(keep, float(keep.sum()) * self.gb, out=host_mask)
                self.maskw.copy_(host_mask, non_blocking=True)
        ev.record()

    def __call__(self, indices):
        """Run one optimiser step on the receivers ``indices``; returns the static loss tensors
        (valid until the next call)."""
        if self.graph_a is None:
            self.capture(indices)
        self._load_inputs(indices)
        self._ensure_records()
        self.graph_a.replay()
        if self.graph_b is not None:
            self.tr._allreduce()
            self.graph_b.replay()
        return self.losses
