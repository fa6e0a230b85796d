"""Differentiable GFDN models on the MI355X hot path.

Interface mirror of the reference's src/diff_gfdn/model.py: DiffGFDN (:24-299),
DiffGFDNVarSourceReceiverPos (:303-452), DiffGFDNVarReceiverPos (:502-661), DiffGFDNSinglePos (:667-969),
DiffDirectionalFDNVarReceiverPos (:975-1126) -- same constructor arguments, same
``forward(x: Dict)`` contract, same parameter / buffer names (state dicts interchange).

What changes is the evaluation: the reference forms P = inv(D Gamma^{-1} - A) for every bin and
contracts (B, N, K) temporaries with einsum; here one per-bin SOLVE y = P b is shared by the
whole batch (b does not depend on the receiver) and the position-dependent part is the
bandwidth-bound output stage  H[b,k] = sum_g gain[b,g] sum_{n in g} c_n y_n[k] + d[b,k],
both hand-written HIP kernels (csrc/solve.hip).  Outputs are complex64 (the reference's are
complex128 only because the complex128 direct path is added to a complex64 result).
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from .config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
from .feedback_loop import FeedbackLoop, decay_times_to_gain_per_sample
from . import hip_ops as ops
from .functional import FrequencyGrid, GroupSums, OutputStage, ResolventSolve, SHOutputStage, SosOutputStage
from .gain_filters import (Directional_Beamforming_Weights_from_MLP, Gains_from_MLP, SVF_from_MLP,
                           svf_cascade_response, svf_cutoff_frequencies)


class DiffGFDN(nn.Module):
    """Parent module (reference model.py:24-299)."""

    def __init__(self, sample_rate: int, num_groups: int, delays: List[int], device: torch.device,
                 feedback_loop_config: FeedbackLoopConfig, use_absorption_filters: bool,
                 learn_common_decay_times: bool, common_decay_times: Optional[List] = None,
                 band_centre_hz: Optional[List] = None, colorless_fdn_params: Optional[List] = None,
                 use_colorless_loss: bool = False, absorption_filter_coeffs: Optional[torch.Tensor] = None):
        """``absorption_filter_coeffs`` (N, S, 3, 2) [SOS, GEQ design] or (N, order, 2) [IIR, Prony design]: the
        per-line absorption filters when ``use_absorption_filters``.  The reference DESIGNS them from the decay
        times at construction (model.py:131-153 -> absorption_filters.py:108-155, an L-BFGS graphic-equaliser
        fit: pre-processing outside this path) and keeps the result in the checkpoint as the ``delay_filters``
        buffer -- that tensor is what is passed here."""
        super().__init__()
        if use_absorption_filters and absorption_filter_coeffs is None:
            raise NotImplementedError("use_absorption_filters: pass absorption_filter_coeffs (the reference's "
                                      "'delay_filters' buffer); the GEQ / Prony filter DESIGN is not part of this path")
        self._absorption_coeffs = absorption_filter_coeffs
        self.sample_rate = sample_rate
        self.device = device
        self.num_groups = num_groups
        self.num_delay_lines = len(delays)
        self.num_delay_lines_per_group = int(self.num_delay_lines / self.num_groups)
        self.use_absorption_filters = use_absorption_filters
        self.band_centre_hz = band_centre_hz
        self.common_decay_times = common_decay_times
        self.learn_common_decay_times = learn_common_decay_times
        self.use_colorless_loss = use_colorless_loss
        n = self.num_delay_lines_per_group
        self.delays_by_group = [torch.tensor(delays[i:i + n]) for i in range(0, self.num_delay_lines, n)]
        self.register_buffer('delay_buffer', torch.tensor(delays, dtype=torch.float32))
        self._ones = self._eye = None
        self._init_io_gains(colorless_fdn_params)
        self._init_absorption()
        self._init_feedback(feedback_loop_config, colorless_fdn_params)

    @property
    def delays(self) -> torch.Tensor:
        return self.delay_buffer

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        if self.gain_per_sample is not None:
            # keep the loop's gains aliased to the checkpointed buffer (load_state_dict copies in place)
            self.feedback_loop.delay_line_gains = self.delay_filters
        return self

    # reference :95-122 -- draws input_gains then output_gains, (2 randn - 1)/N
    def _init_io_gains(self, colorless_fdn_params):
        N = self.num_delay_lines
        if colorless_fdn_params is None:
            self.input_gains = nn.Parameter((2 * torch.randn(N, 1) - 1) / N)
            self.output_gains = nn.Parameter((2 * torch.randn(N, 1) - 1) / N)
        else:
            G = self.num_groups
            self.register_buffer('input_gains', torch.tensor(
                [colorless_fdn_params[i].opt_input_gains.tolist() for i in range(G)]).view(-1, 1),
                persistent=False)
            self.register_buffer('output_gains', torch.tensor(
                [colorless_fdn_params[i].opt_output_gains.tolist() for i in range(G)]).view(-1, 1),
                persistent=False)

    # reference :124-166 (broadband gains branch)
    def _init_absorption(self):
        if self.use_absorption_filters:
            self.gain_per_sample = torch.as_tensor(self._absorption_coeffs, dtype=torch.float32).clone()
            if self.gain_per_sample.shape[0] != self.num_delay_lines:
                raise ValueError("absorption_filter_coeffs: one filter per delay line")
            self.register_buffer('delay_filters', self.gain_per_sample)
            return
        if self.common_decay_times is None or self.learn_common_decay_times:
            self.gain_per_sample = None
            return
        cdt = np.squeeze(self.common_decay_times)
        cdt = np.atleast_1d(cdt)
        vals = [decay_times_to_gain_per_sample(cdt[i], self.delays_by_group[i].numpy(),
                                               self.sample_rate).tolist()
                for i in range(self.num_groups)]
        self.gain_per_sample = torch.flatten(torch.tensor(vals))
        self.register_buffer('delay_filters', self.gain_per_sample)

    # reference :168-207
    def _init_feedback(self, cfg: FeedbackLoopConfig, colorless_fdn_params):
        M0 = None
        if colorless_fdn_params is not None:
            M0 = torch.stack([torch.from_numpy(colorless_fdn_params[i].opt_feedback_matrix)
                              for i in range(self.num_groups)], dim=0).to(torch.float32)
        self.feedback_loop = FeedbackLoop(
            self.sample_rate, self.num_groups, self.num_delay_lines_per_group, self.delay_buffer,
            self.use_absorption_filters, gains=self.gain_per_sample,
            use_zero_coupling=cfg.use_zero_coupling, common_decay_times=self.common_decay_times,
            coupling_matrix_type=cfg.coupling_matrix_type, coupling_matrix_order=cfg.pu_matrix_order,
            colorless_feedback_matrix=M0)

    # -- hot-path pieces ---------------------------------------------------------------------------
    def delay_line_responses(self, z: torch.Tensor, transpose: bool = False,
                             b: Optional[torch.Tensor] = None) -> torch.Tensor:
        """y_n[k] = ((D Gamma^{-1} - A)^{-1} b)_n  -> (K, N) complex64."""
        b = self.input_gains if b is None else b
        return self.feedback_loop.resolvent_apply(z, b, transpose)

    def group_transfer(self, z: torch.Tensor) -> torch.Tensor:
        """T (K, G, G'):  T[k, g, g'] = c_g^T P_{g g'}(z_k) b_{g'}  -- the group-to-group transfer functions every
        frequency-dependent in/out filter multiplies (H = sum_{g,g'} Cout_g T_{g g'} Cin_{g'}).  Zero coupling:
        one solve, diagonal in (g, g'); coupled feedback matrix: one solve per source group."""
        G, n = self.num_groups, self.num_delay_lines_per_group
        fl = self.feedback_loop
        c = self.output_gains.reshape(1, -1)
        if fl.uncoupled:
            S = (self.delay_line_responses(z) * c).reshape(-1, G, n).sum(-1)                    # (K, G)
            return torch.diag_embed(S)
        b = self.input_gains.reshape(G, n)
        cols = []
        for gs in range(G):
            mask = torch.zeros_like(b)
            mask[gs] = 1.0
            Yg = self.delay_line_responses(z, b=(b * mask).reshape(-1, 1))
            cols.append((Yg * c).reshape(-1, G, n).sum(-1))
        return torch.stack(cols, dim=-1)

    def sub_fdn_responses(self, z: torch.Tensor) -> torch.Tensor:
        """Un-damped per-group responses y^(g) = (D - M_g)^{-1} b_g with the RAW parameter M_g
        (reference model.py:237-240) -> (K, N) complex64."""
        grid = FrequencyGrid.of(z)
        M = self.feedback_loop.M
        if self._ones is None or self._ones.device != M.device:
            self._ones = torch.ones(self.num_delay_lines, dtype=torch.float32, device=M.device)
            self._eye = torch.eye(self.num_groups, dtype=torch.float32, device=M.device)
        return ResolventSolve.apply(M, self._ones, self.input_gains.reshape(-1), grid,
                                    self.delay_buffer, False)

    # The group responses of the colorless branch by real transforms of the blocks' coefficient sequences instead of per-bin
    # eliminations (functional.SubFdnTransforms): needs integer delay lengths on the reference's rfftfreq grid and a caller
    # that does not ask for the per-delay-line responses; on for the directional model, whose 9 x 9 eliminations are the
    # largest kernels of its step
    sub_fdn_by_transforms = False

    def sub_fdn_group_sums(self, z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(S (G, K), Ysub (K, N)):  S[g][k] = sum_{n in g} c_n y^(g)_n[k] = Hout[k, g].  (Ysub: None on the transform path)"""
        n = self.num_delay_lines_per_group
        if self.sub_fdn_by_transforms and not self.per_delay_output and n <= 9 and z.is_cuda:
            grid = FrequencyGrid.of(z)
            T_seq = ops.tfp_plan(self.delay_buffer, n, grid.rfft_nfft) if grid.rfft_nfft else None
            if T_seq is not None:
                from .functional import SubFdnTransforms
                M = self.feedback_loop.M
                S = SubFdnTransforms.apply(M.reshape(self.num_groups, n, n), self.input_gains.reshape(-1),
                                           self.output_gains.reshape(-1), self.delay_buffer, n, grid.rfft_nfft, T_seq)
                return S, None
        Ysub = self.sub_fdn_responses(z)
        if ops.group_sums_supported(self.num_groups, self.num_delay_lines_per_group):
            S = GroupSums.apply(Ysub, self.output_gains.reshape(-1), self.num_groups, self.num_delay_lines_per_group)
        else:
            S = OutputStage.apply(Ysub, self.output_gains.reshape(-1), self._eye,
                                  self.num_delay_lines_per_group, None, None)
        return S, Ysub

    # ``Hout_per_del`` (N, K, G) of :meth:`sub_fdn_output` costs five passes over a 42 MB tensor at K = 65 537, N = 27 and
    # no loss of this path reads it: the trainers switch it off for the models they step (the reference returns it
    # always, so it stays the default for everyone else)
    per_delay_output = True

    def sub_fdn_output(self, z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(Hout (K, G), Hout_per_del (N, K, G)) as reference model.py:209-252 (Hout_per_del: None when
        ``per_delay_output`` is off)."""
        S, Ysub = self.sub_fdn_group_sums(z)
        if not self.per_delay_output:
            return S.T, None
        G, n, N = self.num_groups, self.num_delay_lines_per_group, self.num_delay_lines
        K = Ysub.shape[0]
        scaled = (Ysub * self.output_gains.reshape(1, N)).T                 # (N, K): c_n y_n
        Hpd = torch.zeros((N, K, G), dtype=torch.complex64, device=Ysub.device)
        for g in range(G):
            Hpd[g * n:(g + 1) * n, :, g] = scaled[g * n:(g + 1) * n]
        return S.T, Hpd

    @torch.no_grad()
    def get_param_dict(self) -> Dict:
        out = {'delays': self.delay_buffer.squeeze().cpu().numpy(),
               'gains_per_sample': self.feedback_loop.current_gains().squeeze().cpu().numpy(),
               'input_gains': self.input_gains.squeeze().cpu().numpy(),
               'output_gains': self.output_gains.squeeze().cpu().numpy()}
        out.update({k: v for k, v in self.feedback_loop.get_param_dict().items()
                    if k in ('coupled_feedback_matrix', 'individual_mixing_matrix', 'coupling_matrix')})
        return out


class DiffGFDNVarReceiverPos(DiffGFDN):
    """GFDN for a grid of receiver positions, output gains from an MLP (reference :502-661)."""

    def __init__(self, sample_rate: int, num_groups: int, delays: List[int], device: torch.device,
                 feedback_loop_config: FeedbackLoopConfig, output_filter_config: OutputFilterConfig,
                 use_absorption_filters: bool, learn_common_decay_times: Optional[bool] = False,
                 common_decay_times: Optional[List] = None, band_centre_hz: Optional[List] = None,
                 colorless_fdn_params: Optional[List] = None, use_colorless_loss: bool = False,
                 absorption_filter_coeffs: Optional[torch.Tensor] = None):
        super().__init__(sample_rate, num_groups, delays, device, feedback_loop_config,
                         use_absorption_filters, learn_common_decay_times, common_decay_times,
                         band_centre_hz, colorless_fdn_params, use_colorless_loss, absorption_filter_coeffs)
        self.use_svf_in_output = output_filter_config.use_svfs
        self.input_scalars = torch.ones(self.num_groups, 1)
        if self.use_svf_in_output:
            # receiver-position dependent SVF cascades (reference :544-555): the output "gains" become (B, G, K)
            # complex responses; the solve is still shared by the batch, the contraction with the responses is
            # evaluated by torch ops on the device (first version of SURVEY §8 f-2: correct, not yet fused)
            self.output_filters = SVF_from_MLP(
                self.sample_rate, self.num_groups, self.num_delay_lines_per_group,
                output_filter_config.num_fourier_features, output_filter_config.num_hidden_layers,
                output_filter_config.num_neurons_per_layer, output_filter_config.encoding_type,
                output_filter_config.compress_pole_factor)
        else:
            self.output_scalars = Gains_from_MLP(
                self.num_groups, self.num_delay_lines_per_group,
                output_filter_config.num_fourier_features, output_filter_config.num_hidden_layers,
                output_filter_config.num_neurons_per_layer, output_filter_config.encoding_type)

    def forward(self, x: Dict, output_scalars: Optional[torch.Tensor] = None,
                subband_filter: Optional[torch.Tensor] = None):
        """H(z) = c(z)^T (D Gamma^{-1} - A)^{-1} b + d(z)   (reference :569-625).

        ``subband_filter`` (K,) optionally fuses the trainer's H * filter (trainer.py:459) into
        the output stage; default None returns the unfiltered H exactly like the reference."""
        z = x['z_values']
        self.feedback_loop.new_forward()
        self.batch_size = x['listener_position'].shape[0]
        if self.use_svf_in_output:
            # H = sum_g Co[b][g][k] T[k][g] + d[b][k] with the (B, G, K) filter responses evaluated inside the
            # contraction kernel (csrc/svf.hip) from their biquad coefficients
            coef = self.output_filters.biquad_coefficients(x)                            # (B, G, S, 6)
            T = self.group_transfer(z).sum(-1)                                           # (K, G)
            H = SosOutputStage.apply(coef, T.contiguous(), x['target_early_response'], z)
            if subband_filter is not None:
                H = H * subband_filter
            if self.use_colorless_loss:
                return H, self.sub_fdn_output(z)
            return H
        if output_scalars is None:
            rgain = self.output_scalars.group_gains(x)
        else:
            assert output_scalars.shape == (self.batch_size, self.num_groups)
            rgain = output_scalars
        Y = self.delay_line_responses(z)
        H = OutputStage.apply(Y, self.output_gains.reshape(-1), rgain.to(torch.float32),
                              self.num_delay_lines_per_group, x['target_early_response'],
                              subband_filter)
        if self.use_colorless_loss:
            return H, self.sub_fdn_output(z)
        return H

    @torch.no_grad()
    def get_param_dict_inference(self, data: Dict) -> Dict:
        return {'output_scalars': self.output_scalars.get_param_dict(data)['gains']}

    @torch.no_grad()
    def get_param_dict(self) -> Dict:
        out = super().get_param_dict()
        out['input_scalars'] = self.input_scalars.squeeze().cpu().numpy()
        return out


def _scaled_first_section(coef: torch.Tensor, gain: torch.Tensor) -> torch.Tensor:
    """coef (B, G, S, 6) biquad cascades with the first section's numerator times gain (B, G): the cascade's response
    times a real gain, without touching the kernel's interface."""
    scale = torch.ones_like(coef)
    scale[:, :, 0, :3] = gain.to(coef.dtype)[:, :, None]
    return coef * scale


class DiffGFDNVarSourceReceiverPos(DiffGFDN):
    """GFDN for a grid of source AND receiver positions: per-group input gains from the source position and
    output gains from the receiver position, one MLP each (reference model.py:303-452).

    H[b,k] = sum_{g,g'} r[b,g] s[b,g'] c_g^T P_{g g'}(z_k) b_{g'} + d[b,k].  With zero inter-group coupling P is
    block diagonal, the solve y = P b is shared by the batch exactly as in DiffGFDNVarReceiverPos and the two
    gains simply multiply in the output stage; with coupling, one solve per source group (b masked to the
    group) gives the G x G group transfer functions, contracted with the gains by torch ops on the device."""

    def __init__(self, sample_rate: int, num_groups: int, delays: List[int], device: torch.device,
                 feedback_loop_config: FeedbackLoopConfig, output_filter_config: OutputFilterConfig,
                 input_filter_config: OutputFilterConfig, use_absorption_filters: bool,
                 learn_common_decay_times: bool, common_decay_times: Optional[List] = None,
                 band_centre_hz: Optional[List] = None, colorless_fdn_params: Optional[List] = None,
                 use_colorless_loss: bool = False):
        super().__init__(sample_rate, num_groups, delays, device, feedback_loop_config,
                         use_absorption_filters, learn_common_decay_times, common_decay_times,
                         band_centre_hz, colorless_fdn_params, use_colorless_loss)
        # SVF cascades from an MLP on either side (reference :347-400): the side's group factor becomes a (B, G, K) filter
        # response instead of a (B, G) gain -- same attribute names as the reference (output_filters / input_filters)
        self.use_svf_in_output = output_filter_config.use_svfs
        self.use_svf_in_input = input_filter_config.use_svfs
        n = self.num_delay_lines_per_group

        def side(cfg, position_type):
            if cfg.use_svfs:
                return SVF_from_MLP(self.sample_rate, self.num_groups, n, cfg.num_fourier_features, cfg.num_hidden_layers,
                                    cfg.num_neurons_per_layer, cfg.encoding_type, cfg.compress_pole_factor,
                                    position_type=position_type)
            return Gains_from_MLP(self.num_groups, n, cfg.num_fourier_features, cfg.num_hidden_layers,
                                  cfg.num_neurons_per_layer, cfg.encoding_type, position_type=position_type)
        if self.use_svf_in_output:
            self.output_filters = side(output_filter_config, "output_gains")
        else:
            self.output_scalars = side(output_filter_config, "output_gains")
        if self.use_svf_in_input:
            self.input_filters = side(input_filter_config, "input_gains")
        else:
            self.input_scalars = side(input_filter_config, "input_gains")

    def forward(self, x: Dict, subband_filter: Optional[torch.Tensor] = None):
        z = x['z_values']
        self.feedback_loop.new_forward()
        self.batch_size = x['listener_position'].shape[0]
        G, n = self.num_groups, self.num_delay_lines_per_group
        fl = self.feedback_loop
        if self.use_svf_in_output or self.use_svf_in_input:
            return self._forward_filters(x, subband_filter)
        r = self.output_scalars.group_gains(x).to(torch.float32)          # (B, G) from the receiver position
        s = self.input_scalars.group_gains(x).to(torch.float32)           # (B, G) from the source position
        if fl.uncoupled:
            Y = self.delay_line_responses(z)
            H = OutputStage.apply(Y, self.output_gains.reshape(-1), r * s, n, x['target_early_response'],
                                  subband_filter)
        else:
            # T[k, g, g'] = c_g^T P_{g g'} b_{g'}: one solve per source group
            b = self.input_gains.reshape(G, n)
            cols = []
            for gs in range(G):
                mask = torch.zeros_like(b)
                mask[gs] = 1.0
                Yg = self.delay_line_responses(z, b=(b * mask).reshape(-1, 1))          # (K, N)
                cols.append((Yg * self.output_gains.reshape(1, -1)).reshape(-1, G, n).sum(-1))   # (K, G)
            T = torch.stack(cols, dim=-1)                                                # (K, G, G')
            H = torch.einsum('bg,bh,kgh->bk', r.to(T.dtype), s.to(T.dtype), T) + x['target_early_response']
            if subband_filter is not None:
                H = H * subband_filter
        if self.use_colorless_loss:
            return H, self.sub_fdn_output(z)
        return H

    def _forward_filters(self, x: Dict, subband_filter: Optional[torch.Tensor] = None):
        """SVF filters on the input and / or the output side (reference :423-432: C = output_filters(x) C_init,
        B = input_filters(x) B_init): H[b,k] = sum_{g,g'} Fo[b,g,k] Fi[b,g',k] T[k,g,g'] + d[b,k] with the group transfer
        functions T = c_g^T P_{g g'} b_{g'} of the shared per-bin solve.  Zero coupling (T diagonal): the two sides' cascades
        of a group are ONE cascade of up to 22 sections (a side with scalar gains scales the other side's first
        numerator), evaluated inside the contraction kernel (csrc/svf.hip, SosOutputStage) -- the (B, G, K) filter responses
        never exist.  With coupling: the responses are evaluated (sos_cascade_response) and contracted by torch ops."""
        from .gain_filters import sos_cascade_response
        z = x['z_values']
        T = self.group_transfer(z)                                                   # (K, G, G')
        co = self.output_filters.biquad_coefficients(x) if self.use_svf_in_output else None      # (B, G, S, 6)
        ci = self.input_filters.biquad_coefficients(x) if self.use_svf_in_input else None
        r = None if self.use_svf_in_output else self.output_scalars.group_gains(x).to(torch.float32)
        s = None if self.use_svf_in_input else self.input_scalars.group_gains(x).to(torch.float32)
        if self.feedback_loop.uncoupled:
            if co is not None and ci is not None:
                coef = torch.cat((co, ci), dim=2)
            elif co is not None:
                coef = _scaled_first_section(co, s)
            else:
                coef = _scaled_first_section(ci, r)
            H = SosOutputStage.apply(coef.contiguous(), T.sum(-1).contiguous(), x['target_early_response'], z)
        else:
            K = z.shape[-1]
            Fo = sos_cascade_response(z, co) if co is not None else r.to(torch.complex64)[:, :, None].expand(-1, -1, K)
            Fi = sos_cascade_response(z, ci) if ci is not None else s.to(torch.complex64)[:, :, None].expand(-1, -1, K)
            H = torch.einsum('bgk,bhk,kgh->bk', Fo, Fi, T.to(torch.complex64)) + x['target_early_response']
        if subband_filter is not None:
            H = H * subband_filter
        if self.use_colorless_loss:
            return H, self.sub_fdn_output(z)
        return H

    @torch.no_grad()
    def get_param_dict_inference(self, data: Dict) -> Dict:
        out = {}
        for name, svf in (('output', self.use_svf_in_output), ('input', self.use_svf_in_input)):
            if svf:
                prm = getattr(self, f'{name}_filters').get_param_dict(data)
                out[f'{name}_svf_params'], out[f'{name}_biquad_coeffs'] = prm['svf_params'], prm['biquad_coeffs']
            else:
                out[f'{name}_scalars'] = getattr(self, f'{name}_scalars').get_param_dict(data)['gains']
        return out


class DiffGFDNSinglePos(DiffGFDN):
    """GFDN for one source-receiver pair with learnable per-group scalars (reference :667-969)."""

    def __init__(self, sample_rate: int, num_groups: int, delays: List[int], device: torch.device,
                 feedback_loop_config: FeedbackLoopConfig, output_filter_config: OutputFilterConfig,
                 use_absorption_filters: bool, learn_common_decay_times: Optional[bool] = False,
                 common_decay_times: Optional[List] = None, band_centre_hz: Optional[List] = None,
                 colorless_fdn_params: Optional[List] = None, use_colorless_loss: bool = False,
                 input_filter_config: Optional[OutputFilterConfig] = None):
        super().__init__(sample_rate, num_groups, delays, device, feedback_loop_config,
                         use_absorption_filters, learn_common_decay_times, common_decay_times,
                         band_centre_hz, colorless_fdn_params, use_colorless_loss)
        self.use_svf_in_input = input_filter_config.use_svfs if input_filter_config is not None else False
        self.use_svf_in_output = output_filter_config.use_svfs
        G = self.num_groups
        if self.use_svf_in_output or self.use_svf_in_input:
            self.svf_cutoff_freqs = svf_cutoff_frequencies(self.sample_rate)
            self.num_biquads = len(self.svf_cutoff_freqs)
            self.compress_pole_factor = output_filter_config.compress_pole_factor
        # draws in the reference's order: source side first (:723-724), [resonance ~ randn, gain = 0 dB]
        if self.use_svf_in_input:
            init = torch.randn(G, self.num_biquads, 2)
            init[..., 1] = 0.0
            self.input_svf_params = nn.Parameter(init)                       # reference :729-737
        else:
            self.input_scalars = nn.Parameter(torch.ones(G, 1) / np.sqrt(G))     # reference :749-750
        if self.use_svf_in_output:
            init = torch.randn(G, self.num_biquads, 2)
            init[..., 1] = 0.0
            self.output_svf_params = nn.Parameter(init)                      # reference :756-764
        else:
            self.output_scalars = nn.Parameter(torch.ones(G, 1) / np.sqrt(G))    # reference :776-777

    def forward(self, x: Dict):
        """reference :779-836; inputs are (K,) tensors, output H is (K,)."""
        z = x['z_values']
        self.feedback_loop.new_forward()
        n = self.num_delay_lines_per_group
        if self.use_svf_in_input or self.use_svf_in_output:
            # per-group SVF cascades on either side (reference get_filter :838-911): H = sum_{g,g'} Co_g T_{g g'} Ci_{g'}
            K = z.numel()
            T = self.group_transfer(z)                                                   # (K, G, G')
            Co = (svf_cascade_response(z, self.svf_cutoff_freqs, self.output_svf_params, self.compress_pole_factor)
                  if self.use_svf_in_output else self.output_scalars.to(torch.complex64).expand(-1, K))
            Ci = (svf_cascade_response(z, self.svf_cutoff_freqs, self.input_svf_params, self.compress_pole_factor)
                  if self.use_svf_in_input else self.input_scalars.to(torch.complex64).expand(-1, K))
            H = torch.einsum('gk,kgh,hk->k', Co, T, Ci) + x['target_early_response'].reshape(-1)
            if self.use_colorless_loss:
                return H, self.sub_fdn_output(z)
            return H
        b = self.input_scalars.repeat_interleave(n, dim=0) * self.input_gains
        Y = self.delay_line_responses(z, b=b)
        H = OutputStage.apply(Y, self.output_gains.reshape(-1), self.output_scalars.reshape(1, -1),
                              n, x['target_early_response'].reshape(1, -1), None)
        H = H.reshape(-1)
        if self.use_colorless_loss:
            return H, self.sub_fdn_output(z)
        return H

    @torch.no_grad()
    def get_param_dict(self) -> Dict:
        out = super().get_param_dict()
        out['absorption_coeffs'] = self.feedback_loop.current_gains()
        out['output_scalars'] = self.output_scalars.squeeze().cpu().numpy()
        out['input_scalars'] = self.input_scalars.squeeze().cpu().numpy()
        return out


class DiffDirectionalFDNVarReceiverPos(DiffGFDN):
    """Directional FDN with SH-domain output gains (reference :975-1126)."""

    sub_fdn_by_transforms = True

    def __init__(self, sample_rate: int, num_groups: int, delays: List[int], device: torch.device,
                 feedback_loop_config: FeedbackLoopConfig, output_filter_config: OutputFilterConfig,
                 ambi_order: int, desired_directions=None, use_absorption_filters: bool = False,
                 learn_common_decay_times: Optional[bool] = False,
                 common_decay_times: Optional[List] = None, band_centre_hz: Optional[List] = None,
                 colorless_fdn_params: Optional[List] = None, use_colorless_loss: bool = False,
                 analysis_matrix=None):
        super().__init__(sample_rate, num_groups, delays, device, feedback_loop_config,
                         use_absorption_filters, learn_common_decay_times, common_decay_times,
                         band_centre_hz, colorless_fdn_params, use_colorless_loss)
        self.ambi_order = ambi_order
        assert self.num_delay_lines_per_group == (ambi_order + 1) ** 2, \
            "Number of delay lines per group must be equal to the number of ambisonics channels"
        self.input_scalars = torch.ones(self.num_groups, 1)
        self.use_svf_in_output = False
        self.sh_output_scalars = Directional_Beamforming_Weights_from_MLP(
            self.num_groups, ambi_order, output_filter_config.num_fourier_features,
            output_filter_config.num_hidden_layers, output_filter_config.num_neurons_per_layer,
            desired_directions=desired_directions,
            beamformer_type=output_filter_config.beamformer_type,
            use_skip_connections=output_filter_config.use_skip_connections,
            analysis_matrix=analysis_matrix)

    def forward(self, x: Dict, subband_filter: Optional[torch.Tensor] = None):
        """H_sh (B, (order+1)^2, K)   (reference :1043-1094).  NB the reference contracts the FIRST
        index of P with b (einsum 'knm,bnk->bmk'), i.e. P^T b -> transpose solve."""
        z = x['z_values']
        Y, c, w = self.output_stage_inputs(x)
        H = SHOutputStage.apply(Y, c, w, self.num_groups, self.num_delay_lines_per_group, subband_filter)
        if self.use_colorless_loss:
            return H, self.sub_fdn_output(z)
        return H

    def output_stage_inputs(self, x: Dict):
        """(Y (K, N) delay-line responses of the transposed solve, c (N,) output gains, w (B, N) SH weights): what the SH
        output stage (reference :1056-1088) is linear in -- the trainer's time-domain output stage starts from these."""
        self.feedback_loop.new_forward()
        self.batch_size = x['listener_position'].shape[0]
        w = self.sh_output_scalars(x, normalise_weights=True)
        Y = self.delay_line_responses(x['z_values'], transpose=True)
        return Y, self.output_gains.reshape(-1), w.to(torch.float32)

    @torch.no_grad()
    def get_param_dict_inference(self, data: Dict, normalise_weights: bool = False) -> Dict:
        return {'output_scalars':
                self.sh_output_scalars.get_param_dict(data, normalise_weights)['beamformer_weights']}
