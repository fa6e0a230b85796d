"""Receiver-position dependent output gains (reference src/diff_gfdn/gain_filters.py:436-555 and
src/spatial_sampling/model.py:17-190).  Small MLPs on PyTorch; what they feed -- the output
stage over K bins -- runs in the HIP kernels and takes the (B, G) gains directly instead of the
reference's (B, N, K) repeated tensor."""
import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import nn

from .config import BeamformerType, FeatureEncodingType
from . import hip_ops as ops
from .dnn import MLP, MLP_SkipConnections, ScaledSigmoid, SinusoidalEncoding


def svf_cutoff_frequencies(sample_rate: float) -> torch.Tensor:
    """Normalised SVF cut-offs pi f / fs of the 11-section equaliser: low shelf at 62.5 / sqrt 2 Hz, peaking
    sections at the octaves 62.5 ... 16 000 Hz, high shelf at 16 000 sqrt 2 Hz (reference filters/geq.py:9-56
    ``eq_freqs`` with its defaults, gain_filters.py:299-303 -- no tangent pre-warping, as there)."""
    centre, c = [], 31.25
    while c < 16000:
        c = c * 2.0
        centre.append(c)
    f = [centre[0] / np.power(2, 0.5)] + centre + [centre[-1] * np.power(2, 0.5)]
    return torch.pi * torch.tensor(f) / sample_rate


def svf_biquad_coefficients(cutoff: torch.Tensor, raw_params: torch.Tensor,
                            compress_pole_factor: float = 1.0) -> torch.Tensor:
    """(..., S, 6) float32 biquad coefficients [b0 b1 b2 a0 a1 a2] of cascades of state-variable-filter sections.

    raw_params (..., S, 2): unconstrained [resonance, gain dB] per section, mapped through the reference's scaled
    sigmoids (resonance in (1e-6, 1), gain in (-6, 6) dB; gain_filters.py:327-330, model.py:733-737); section 0 is
    a low shelf, the last a high shelf, the others peaking (gain_filters.py:372-380); SVF -> biquad as
    ``BiquadCascade.from_svf_coeffs`` (:117-151, mixing coefficients of ``SVF.__post_init__`` :36-103).  Formed as
    the reference does -- float64 cut-offs times float32 parameters -- and stored as float32.  Device tensors go
    through one HIP launch each way (csrc/svf.hip); the torch expression below is the CPU / reference form."""
    if raw_params.is_cuda and raw_params.dtype == torch.float32:
        from .functional import SvfCoefficients
        return SvfCoefficients.apply(raw_params.contiguous(), cutoff.to(raw_params.device).to(torch.float64),
                                     1.0 if compress_pole_factor is None else float(compress_pole_factor))
    dev = raw_params.device
    S = raw_params.shape[-2]
    R = 1e-6 + (1.0 - 1e-6) * torch.sigmoid(raw_params[..., 0])
    G = torch.pow(10.0, (-6.0 + 12.0 * torch.sigmoid(raw_params[..., 1])) * 0.05)      # db2lin
    f = cutoff.to(dev).to(torch.float64)
    R, G = R.to(torch.float64), G.to(torch.float64)
    idx = torch.arange(S, device=dev)
    low, high = (idx == 0), (idx == S - 1)
    one = torch.ones_like(G)
    m_lp = torch.where(low, G, one)
    m_hp = torch.where(high, G, one)
    m_bp = torch.where(low | high, 2 * R * torch.sqrt(G), 2 * R * G)
    cpf = compress_pole_factor
    b0 = f ** 2 * m_lp + f * m_bp + m_hp
    b1 = (2 * f ** 2 * m_lp - 2 * m_hp) * cpf
    b2 = (f ** 2 * m_lp - f * m_bp + m_hp) * cpf ** 2
    a0 = f ** 2 + 2 * R * f + 1
    a1 = (2 * f ** 2 - 2) * cpf + torch.zeros_like(R)
    a2 = (f ** 2 - 2 * R * f + 1) * cpf ** 2
    return torch.stack([b0, b1, b2, a0, a1, a2], dim=-1).to(torch.float32)


def sos_cascade_response(z: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
    """Frequency response (..., K) complex64 of the cascades coef (..., S, 6), as ``SOSFilter.forward``
    (gain_filters.py:221-241).  The sections are evaluated in complex128 (b0 + b1 z^-1 + b2 z^-2 cancels to O(f^2)
    at low frequencies: in complex64 the shelves lose 3 digits there), the running product is kept in complex64 like
    the reference.  Device tensors without a gradient go through the HIP kernel (csrc/svf.hip: same arithmetic, no
    (..., K) intermediates); gradients flow through the torch expression here or, for the receiver-dependent output
    filters of the grid model, through :class:`~diffgfdn_amd.functional.SosOutputStage`."""
    if coef.is_cuda and not (coef.requires_grad and torch.is_grad_enabled()):
        lead = coef.shape[:-2]
        out = ops.sos_response(coef.reshape(-1, coef.shape[-2], 6), z.to(coef.device))
        return out.reshape(*lead, z.numel())
    zi = 1.0 / z.to(torch.complex128)
    zi2 = zi * zi
    H = None
    for k in range(coef.shape[-2]):
        c = lambda j: coef[..., k, j].to(torch.float64).unsqueeze(-1)
        sec = ((c(0) + c(1) * zi + c(2) * zi2) / (c(3) + c(4) * zi + c(5) * zi2)).to(torch.complex64)
        H = sec if H is None else H * sec
    return H


def svf_cascade_response(z: torch.Tensor, cutoff: torch.Tensor, raw_params: torch.Tensor,
                         compress_pole_factor: float = 1.0) -> torch.Tensor:
    """Frequency response (..., K) complex64 of cascades of state-variable-filter sections:
    :func:`svf_biquad_coefficients` then :func:`sos_cascade_response`."""
    return sos_cascade_response(z, svf_biquad_coefficients(cutoff, raw_params, compress_pole_factor))


class SVF_from_MLP(nn.Module):
    """Receiver-position dependent SVF output (or source-position dependent input) filters: per (position,
    group) a cascade of 11 state-variable sections whose [resonance, gain] come from an MLP (reference
    gain_filters.py:262-402).  ``forward`` returns the reference's (B, N, K) tensor; the models use
    :meth:`group_responses` (B, G, K)."""

    def __init__(self, sample_rate: float, num_groups: int, num_delay_lines_per_group: int,
                 num_fourier_features: int, num_hidden_layers: int, num_neurons: int,
                 encoding_type: FeatureEncodingType = FeatureEncodingType.SINE,
                 compress_pole_factor: Optional[float] = 1.0, position_type: str = "output_gains",
                 device: Optional[torch.device] = 'cpu'):
        super().__init__()
        if encoding_type != FeatureEncodingType.SINE:
            raise NotImplementedError("only sinusoidal encoding is on the accelerated path")
        self.num_groups = num_groups
        self.num_delay_lines_per_group = num_delay_lines_per_group
        self.num_delay_lines = num_groups * num_delay_lines_per_group
        self.position_type = position_type
        self.encoding_type = encoding_type
        self.compress_pole_factor = compress_pole_factor
        self.device = device
        # non-persistent buffer: follows .to(device) (no host-to-device copy inside a captured step), no state-dict key
        self.register_buffer('svf_cutoff_freqs', svf_cutoff_frequencies(sample_rate), persistent=False)
        self.num_biquads = len(self.svf_cutoff_freqs)
        self.encoder = SinusoidalEncoding(num_fourier_features)
        self._freq_pi = None
        self.mlp = MLP(3 * num_fourier_features * 2, num_hidden_layers, num_neurons, num_groups,
                       self.num_biquads, num_params=2)

    def _fused_ok(self, position: torch.Tensor) -> bool:
        lin = [m for m in self.mlp.model if isinstance(m, nn.Linear)]
        H = lin[0].out_features
        return (position.is_cuda and position.shape[-1] == 3 and lin[0].weight.is_cuda
                and lin[0].weight.dtype == torch.float32 and H <= 256 and lin[-1].out_features <= 256
                and all(m.out_features == H for m in lin[:-1]))

    def raw_parameters(self, x: Dict) -> torch.Tensor:
        """(B, G, S, 2) unconstrained [resonance, gain] per section from the network.  On the GPU encoding and MLP
        run as the fused kernel of csrc/mlp.hip (no output activation); the torch modules are the same network."""
        # NB the reference feeds the RAW listener position here (:340-342), not the normalised one
        position = x['listener_position'] if self.position_type == "output_gains" else x['source_position']
        w = self.mlp.model[0].weight
        position = position.to(w.device)
        if self._fused_ok(position):
            from .functional import MlpGains
            lin = [m for m in self.mlp.model if isinstance(m, nn.Linear)]
            if self._freq_pi is None or self._freq_pi.device != position.device:
                n = self.encoder.num_fourier_features
                f = torch.exp(torch.linspace(math.log(1.0), math.log(32.0), n, device=position.device))
                self._freq_pi = (f * math.pi).contiguous()
            params = [p for m in self.mlp.model for p in m.parameters()]
            raw = MlpGains.apply(position.to(torch.float64).contiguous(), None, self._freq_pi, lin[0].out_features,
                                 len(lin) - 2, lin[-1].out_features, 0.0, 0.0, *params)
            return raw.view(position.shape[0], self.num_groups, self.num_biquads, 2)
        return self.mlp(self.encoder(position).to(w.dtype))

    def biquad_coefficients(self, x: Dict) -> torch.Tensor:
        """(B, G, S, 6) float32 biquad coefficients of the cascades at the batch's positions."""
        self.svf_params = self.raw_parameters(x)                                # (B, G, S, 2) raw
        return svf_biquad_coefficients(self.svf_cutoff_freqs, self.svf_params, self.compress_pole_factor)

    def group_responses(self, x: Dict) -> torch.Tensor:
        w = self.mlp.model[0].weight
        return sos_cascade_response(x['z_values'].to(w.device), self.biquad_coefficients(x))

    @torch.no_grad()
    def get_param_dict(self, x: Dict) -> Dict:
        """{'svf_params': (B, G, S, 2) constrained [resonance, gain dB], 'biquad_coeffs': (B, G, S, 6)} as numpy arrays
        (reference gain_filters.py:404-433: what its inference scripts save per position)."""
        coef = self.biquad_coefficients(x)
        raw = self.svf_params
        prm = torch.stack((torch.sigmoid(raw[..., 0]) * (1.0 - 1e-6) + 1e-6, torch.sigmoid(raw[..., 1]) * 12.0 - 6.0), dim=-1)
        return {'svf_params': prm.squeeze().cpu().numpy(), 'biquad_coeffs': coef.squeeze().cpu().numpy()}

    def forward(self, x: Dict) -> torch.Tensor:
        return self.group_responses(x).repeat_interleave(self.num_delay_lines_per_group, dim=1)


class Gains_from_MLP(nn.Module):
    """Scalar gain per (receiver, group) in (-1, 1) from the normalised receiver position."""

    def __init__(self, num_groups: int, num_delay_lines_per_group: int, num_fourier_features: int,
                 num_hidden_layers: int, num_neurons: int,
                 encoding_type: FeatureEncodingType = FeatureEncodingType.SINE,
                 position_type: str = "output_gains", device: Optional[torch.device] = 'cpu',
                 gain_limits: Optional[Tuple] = None):
        super().__init__()
        if encoding_type != FeatureEncodingType.SINE:
            raise NotImplementedError("only sinusoidal encoding is on the accelerated path")
        self.num_groups = num_groups
        self.num_delay_lines_per_group = num_delay_lines_per_group
        self.position_type = position_type
        self.encoding_type = encoding_type
        self.device = device
        self.encoder = SinusoidalEncoding(num_fourier_features)
        self.mlp = MLP(3 * num_fourier_features * 2, num_hidden_layers, num_neurons, num_groups,
                       num_biquads_in_cascade=1, num_params=1)
        lo, hi = (-1.0, 1.0) if gain_limits is None else gain_limits
        self.scaled_sigmoid = ScaledSigmoid(lower_limit=lo, upper_limit=hi)
        self._freq_pi = None

    def _fused_ok(self, position: torch.Tensor) -> bool:
        lin = [m for m in self.mlp.model if isinstance(m, nn.Linear)]
        H = lin[0].out_features
        return (position.is_cuda and position.shape[-1] == 3 and lin[0].weight.is_cuda
                and lin[0].weight.dtype == torch.float32 and H <= 256 and self.num_groups <= 256
                and all(m.out_features == H for m in lin[:-1]))

    def group_gains(self, x: Dict) -> torch.Tensor:
        """(B, G) gains -- what the HIP output stage consumes.  On the GPU the encoding, the MLP and
        the sigmoid run as one fused kernel (csrc/mlp.hip); the torch modules below are the same
        network (their parameters ARE the kernel's weights) and serve as the CPU / odd-shape path."""
        position = x['norm_listener_position'] if self.position_type == "output_gains" \
            else x['source_position']
        # 'row_index' (MultiRIRDataset.collate(..., lean="rows")): positions are the store of ALL
        # receivers and the batch is the int64 index into it
        rows = x.get('row_index') if self.position_type == "output_gains" else None
        if rows is not None and not self._fused_ok(position):
            position, rows = position[rows], None
        if self._fused_ok(position):
            from .functional import MlpGains
            lin = [m for m in self.mlp.model if isinstance(m, nn.Linear)]
            if self._freq_pi is None or self._freq_pi.device != position.device:
                n = self.encoder.num_fourier_features
                f = torch.exp(torch.linspace(math.log(1.0), math.log(32.0), n, device=position.device))
                self._freq_pi = (f * math.pi).contiguous()
            params = [p for m in self.mlp.model for p in m.parameters()]
            self.gains = MlpGains.apply(position, rows, self._freq_pi, lin[0].out_features, len(lin) - 2,
                                        self.num_groups, self.scaled_sigmoid.lower_limit,
                                        self.scaled_sigmoid.upper_limit, *params)
            return self.gains
        w = self.mlp.model[0].weight
        enc = self.encoder(position.to(w.device))
        raw = self.mlp(enc.to(w.dtype))
        self.gains = self.scaled_sigmoid(raw.view(-1)).view(position.shape[0], self.num_groups)
        return self.gains

    def forward(self, x: Dict) -> torch.Tensor:
        """Reference-shaped output (B, N, K) (gain_filters.py:526-534); kept for API parity, the
        models call :meth:`group_gains`."""
        g = self.group_gains(x)
        return g.repeat_interleave(self.num_delay_lines_per_group, dim=1).unsqueeze(-1).repeat(
            1, 1, len(x['z_values']))

    def get_parameters(self):
        return self.gains

    @torch.no_grad()
    def get_param_dict(self, x: Dict) -> Dict:
        self.group_gains(x)
        return {'gains': self.gains.squeeze().cpu().numpy()}


class Directional_Beamforming_Weights_from_MLP(nn.Module):
    """SH-domain receiver weights (B, G, (order+1)^2) (reference spatial_sampling/model.py:117-190).

    The analysis matrix (J x (order+1)^2) comes from spaudiopy in the reference
    (model.py:52-76), which is not part of this build: pass it as ``analysis_matrix``."""

    def __init__(self, num_groups: int, ambi_order: int, num_fourier_features: int,
                 num_hidden_layers: int, num_neurons: int, desired_directions=None,
                 device: Optional[torch.device] = 'cpu',
                 beamformer_type: Optional[BeamformerType] = None,
                 use_skip_connections: Optional[bool] = False,
                 analysis_matrix: Optional[np.ndarray] = None):
        super().__init__()
        self.num_groups = num_groups
        self.device = device
        self.ambi_order = ambi_order
        self.num_fourier_features = num_fourier_features
        self.num_out_features = (ambi_order + 1) ** 2
        if analysis_matrix is None:
            analysis_matrix = self._design_filterbank(beamformer_type, desired_directions)
        self.register_buffer('analysis_matrix',
                             torch.as_tensor(np.asarray(analysis_matrix), dtype=torch.float32),
                             persistent=False)
        self.encoder = SinusoidalEncoding(num_fourier_features)
        cls = MLP_SkipConnections if use_skip_connections else MLP
        self.mlp = cls(3 * num_fourier_features * 2, num_hidden_layers, num_neurons, num_groups,
                       num_biquads_in_cascade=1, num_params=self.num_out_features)

    def _design_filterbank(self, beamformer_type, desired_directions):
        try:
            import spaudiopy as sp   # optional third-party dependency of the reference
        except ImportError as exc:
            raise RuntimeError("spaudiopy is not installed: pass analysis_matrix explicitly") from exc
        if beamformer_type == BeamformerType.MAX_DI:
            mw = sp.sph.cardioid_modal_weights(self.ambi_order)
        elif beamformer_type == BeamformerType.MAX_RE:
            mw = sp.sph.maxre_modal_weights(self.ambi_order)
        elif beamformer_type == BeamformerType.BUTTER:
            mw = sp.sph.butterworth_modal_weights(self.ambi_order, k=5, n_c=3)
        else:
            mw = np.ones(self.ambi_order + 1)
        A, _ = sp.sph.design_sph_filterbank(self.ambi_order, desired_directions[0, :],
                                            np.pi / 2 - desired_directions[1, :], mw,
                                            mode='energy', sh_type='real')
        return A

    @staticmethod
    def normalise_weights(weights: torch.Tensor) -> torch.Tensor:
        if weights.is_cuda and weights.dtype == torch.float32:
            from .functional import RowNormalise
            return RowNormalise.apply(weights.contiguous())
        return weights / (torch.norm(weights, dim=-1, keepdim=True) + 1e-6)

    def _fused_ok(self, position: torch.Tensor) -> bool:
        if not isinstance(self.mlp, MLP) or isinstance(self.mlp, MLP_SkipConnections):
            return False
        lin = [m for m in self.mlp.model if isinstance(m, nn.Linear)]
        H = lin[0].out_features
        return (position.is_cuda and position.shape[-1] == 3 and lin[0].weight.is_cuda
                and lin[0].weight.dtype == torch.float32 and H <= 256
                and self.num_groups * self.num_out_features <= 256 and all(m.out_features == H for m in lin[:-1]))

    def forward(self, x: Dict, normalise_weights: bool = False) -> torch.Tensor:
        position = x['norm_listener_position']
        w0 = next(self.mlp.parameters())
        if self._fused_ok(position):
            # encoding + [Linear, LayerNorm, ReLU] stack + output Linear as ONE launch each way (csrc/mlp.hip, no
            # output activation) instead of ~30 + ~60 torch launches
            from .functional import MlpGains
            lin = [m for m in self.mlp.model if isinstance(m, nn.Linear)]
            if getattr(self, '_freq_pi', None) is None or self._freq_pi.device != position.device:
                f = torch.exp(torch.linspace(math.log(1.0), math.log(32.0), self.num_fourier_features,
                                             device=position.device))
                self._freq_pi = (f * math.pi).contiguous()
            params = [p for m in self.mlp.model for p in m.parameters()]
            raw = MlpGains.apply(position, None, self._freq_pi, lin[0].out_features, len(lin) - 2,
                                 self.num_groups * self.num_out_features, 0.0, 0.0, *params)
            self.weights = raw.reshape(position.shape[0], self.num_groups, self.num_out_features)
        else:
            enc = self.encoder(position.to(w0.device))
            self.weights = self.mlp(enc.to(w0.dtype)).reshape(position.shape[0], self.num_groups,
                                                              self.num_out_features)
        if normalise_weights:
            self.weights = self.normalise_weights(self.weights)
        return self.weights

    def get_directional_amplitudes(self) -> torch.Tensor:
        out = torch.einsum('jn, bkn-> bjk', self.analysis_matrix, self.weights)
        return 1.0 / (1 + torch.exp(-out))

    def get_parameters(self):
        return self.weights

    @torch.no_grad()
    def get_param_dict(self, x: Dict, normalise_weights: bool = False) -> Dict:
        self.forward(x, normalise_weights=normalise_weights)
        return {'beamformer_weights': self.weights.squeeze().cpu().numpy(),
                'directional_weights': self.get_directional_amplitudes().squeeze().cpu().numpy()}
