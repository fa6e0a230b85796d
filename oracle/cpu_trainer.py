"""CPU restatement of one optimiser step of the reference grid trainer -- TEST INFRASTRUCTURE.

normalize (trainer.py:317-332) + train_step (trainer.py:452-477) with Adam parameter groups
(trainer.py:152-228), built on oracle/gfdn_oracle.py.  Used by tests (F4 fixture: post-Adam state)
and by bench.py's ``cpu_baseline`` leg (kind "port": the reference itself cannot travel to the
GPU box).  Never imported by the product package."""
from typing import Dict, Optional

import numpy as np
import torch

from . import gfdn_oracle as orc


class OracleGridTrainer:
    def __init__(self, params: orc.GridModelParams, *, lr=1e-3, io_lr=1e-2, coupling_angle_lr=1e-2,
                 edr_weight=1.0, edc_weight=1.0, spectral_weight=1.0, sparsity_weight=1.0,
                 use_asym=False, win=4096, hop=2048, subband_filter: Optional[torch.Tensor] = None,
                 learn_alpha: bool = False):
        self.p = params
        self.w = dict(edr=edr_weight, edc=edc_weight, spec=spectral_weight, spars=sparsity_weight)
        self.use_asym = use_asym
        self.win, self.hop = win, hop
        self.subband_filter = subband_filter
        p = params
        for t in (p.input_gains, p.output_gains, p.M):
            t.requires_grad_(True)
        mlp = [t for pair in list(p.mlp_weights) + list(p.mlp_norms) for t in pair]
        for t in mlp:
            t.requires_grad_(True)
        groups = []
        if learn_alpha:
            p.alpha.requires_grad_(True)
            groups.append({'params': [p.alpha], 'lr': coupling_angle_lr})
        groups += [{'params': [p.output_gains], 'lr': io_lr}, {'params': [p.input_gains], 'lr': io_lr},
                   {'params': mlp, 'lr': io_lr},          # names contain 'output_scalars'
                   {'params': [p.M], 'lr': lr}]
        self.optimizer = torch.optim.Adam(groups)

    def losses(self, batch: Dict, edc_mask: Optional[torch.Tensor] = None) -> Dict:
        p = self.p
        H, Hs = orc.grid_model_forward(p, batch)
        if self.subband_filter is not None:
            H = H * self.subband_filter
        fs = p.sample_rate
        tgt = batch['target_rir_response']
        max_samps = orc.ms_to_samps(float(np.max(p.common_decay_times)) * 1e3, fs)
        out = {'edc_loss': self.w['edc'] * orc.edc_loss(tgt, H, max_samps, orc.ms_to_samps(20.0, fs), edc_mask),
               'edr_loss': self.w['edr'] * orc.edr_loss(tgt, H, self.win, self.hop)}
        crit = orc.amse_loss if self.use_asym else orc.mse_loss
        spec = 0.0
        for k in range(p.num_groups):
            hk = Hs[0][..., k]
            spec = spec + self.w['spec'] * crit(hk, torch.ones_like(hk))
            spars = self.w['spars'] * orc.sparsity_loss(orc.ortho_param(p.M[k]))
        out['spectral_loss'] = spec
        out['sparsity_loss'] = spars
        return out

    def normalize(self, batch: Dict):
        with torch.no_grad():
            _, Hs = orc.grid_model_forward(self.p, batch)
        orc.normalize_io_gains(self.p, Hs)

    def train_step(self, batch: Dict, edc_mask: Optional[torch.Tensor] = None):
        self.optimizer.zero_grad()
        losses = self.losses(batch, edc_mask)
        total = sum(losses.values())
        total.backward()
        self.optimizer.step()
        return total.item(), {k: float(v.detach()) for k, v in losses.items()}



def directional_band_step(state: Dict[str, torch.Tensor], delays, analysis_matrix, z, norm_pos, amps, envelopes,
                          num_groups: int, n_per_group: int, num_fourier_features: int, mix_samps: int, edc_samps: int,
                          edc_weight: float = 1.0, colorless: Optional[Dict] = None):
    """Forward, directional EDC loss and backward of ONE band-step of the directional model on the CPU (test
    infrastructure / bench cpu_baseline; reference model.py:1043-1094, trainer.py:853-865, losses.py:333-371): the
    reference's arithmetic under autograd -- complex128 resolvent by torch.linalg.inv, float64 elsewhere.
    ``state``: the module's state dict (CPU tensors).  Returns (loss, H_sh, H_dir, {parameter name: gradient}).
    ``colorless`` = {'spectral_weight', 'sparsity_weight', 'use_asym'}: the colorless terms of the trainer's step on the raw
    sub-FDNs are added to the total (trainer.py:298-313: spectral term per group, sparsity term of the LAST group only, as
    the reference's loop leaves it); the return value then carries a fifth entry, the dict of the three terms."""
    sd = {k: v.detach().cpu().clone() for k, v in state.items()}
    prm = {k: sd[k].clone().requires_grad_(True) for k in ('input_gains', 'output_gains', 'feedback_loop.M')}
    root = 'sh_output_scalars.mlp.model.'
    idx = sorted({int(k[len(root):].split('.')[0]) for k in sd if k.startswith(root)})
    lin, norm = [], []
    for i in idx:
        w = sd[f'{root}{i}.weight'].clone().requires_grad_(True)
        b_ = sd[f'{root}{i}.bias'].clone().requires_grad_(True)
        (lin if w.ndim == 2 else norm).append((w, b_))
        prm[f'{root}{i}.weight'], prm[f'{root}{i}.bias'] = w, b_
    dl = torch.as_tensor(delays, dtype=torch.float32)
    Amat = orc.coupled_feedback_matrix(prm['feedback_loop.M'], sd['feedback_loop.alpha'])
    P = orc.feedback_loop_forward(z, dl, sd['delay_filters'], Amat)
    enc = orc.sinusoidal_encoding(norm_pos, num_fourier_features)
    w_sh = orc.normalise_sh_weights(orc.mlp_forward(enc, lin, norm).reshape(-1, num_groups, n_per_group))
    H_sh = orc.directional_forward(z, prm['input_gains'], prm['output_gains'], w_sh, P, num_groups, n_per_group)
    H_dir = orc.sh_to_directional(torch.as_tensor(analysis_matrix), H_sh)
    loss = edc_weight * orc.directional_edc_loss(H_dir, amps, envelopes, mix_samps, edc_samps)
    if colorless is None:
        loss.backward()
        return loss.detach(), H_sh.detach(), H_dir.detach(), {k: v.grad for k, v in prm.items() if v.grad is not None}
    Hs, _ = orc.sub_fdn_output(z, prm['feedback_loop.M'], prm['input_gains'], prm['output_gains'], dl)
    crit = orc.amse_loss if colorless.get('use_asym', False) else orc.mse_loss
    spec = 0.0
    for k in range(num_groups):
        hk = Hs[..., k]
        spec = spec + colorless['spectral_weight'] * crit(hk, torch.ones_like(hk))
        spars = colorless['sparsity_weight'] * orc.sparsity_loss(orc.ortho_param(prm['feedback_loop.M'][k]))
    terms = {'edc_loss': loss.detach(), 'spectral_loss': spec.detach(), 'sparsity_loss': spars.detach()}
    total = loss + spec + spars
    total.backward()
    return (total.detach(), H_sh.detach(), H_dir.detach(), {k: v.grad for k, v in prm.items() if v.grad is not None},
            terms)
