"""Position -> gain networks: the module tree that holds the parameters (and the CPU / odd-shape path).  On the GPU the
networks of this path run as ONE fused launch each way (csrc/mlp.hip: encoding + [Linear, LayerNorm, ReLU] stack +
output layer [+ sigmoid]) through gain_filters.Gains_from_MLP / SVF_from_MLP / Directional_Beamforming_Weights_from_MLP,
whose kernels read these modules' parameters in place; skip-connection networks stay on torch / hipBLASLt.

Mirrors the module tree and state-dict keys of the reference's src/diff_gfdn/dnn.py
(SinusoidalEncoding :89-126, ScaledSigmoid :21-36, MLP :331-400, MLP_SkipConnections :284-328)
so that checkpoints interchange.  Construction order of the layers matches the reference so a
seeded initialisation draws the same numbers.
"""
import math

import torch
from torch import nn


class ScaledSigmoid(nn.Module):
    """lower + (upper - lower) * sigmoid(x)   (reference dnn.py:21-36)."""

    def __init__(self, lower_limit: float, upper_limit: float):
        super().__init__()
        self.lower_limit = lower_limit
        self.upper_limit = upper_limit

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.lower_limit + (self.upper_limit - self.lower_limit) * (1.0 / (1 + torch.exp(-x)))


class SinusoidalEncoding(nn.Module):
    """(P, F) coordinates -> (P, 2 F num_fourier_features) float32 features, per frequency
    [sin(f pi x) | cos(f pi x)] with f log-spaced in [1, 32]   (reference dnn.py:89-126)."""

    def __init__(self, num_fourier_features: int):
        super().__init__()
        self.num_fourier_features = num_fourier_features

    def forward(self, pos_coords: torch.Tensor) -> torch.Tensor:
        n = self.num_fourier_features
        freqs = torch.exp(torch.linspace(math.log(1.0), math.log(32.0), n, device=pos_coords.device))
        arg = freqs.view(n, 1, 1) * math.pi * pos_coords.unsqueeze(0)          # (n, P, F)
        feats = torch.cat((torch.sin(arg), torch.cos(arg)), dim=-1)            # (n, P, 2F)
        return feats.permute(1, 0, 2).reshape(pos_coords.shape[0], -1).to(torch.float32)


def _he_init(module: nn.Module):
    for layer in module.modules():
        if isinstance(layer, nn.Linear):
            nn.init.kaiming_uniform_(layer.weight, nonlinearity='relu')
            if layer.bias is not None:
                nn.init.constant_(layer.bias, 0)


class MLP(nn.Module):
    """[Linear, LayerNorm, ReLU] x (1 + hidden) + Linear, as ``self.model`` (reference dnn.py:331-400)."""

    def __init__(self, num_pos_features: int, num_hidden_layers: int, num_neurons: int,
                 num_groups: int, num_biquads_in_cascade: int, num_params: int):
        super().__init__()
        self.num_biquads = num_biquads_in_cascade
        self.num_groups = num_groups
        self.num_params = num_params
        out = num_groups * num_params * num_biquads_in_cascade
        layers = [nn.Linear(num_pos_features, num_neurons), nn.LayerNorm(num_neurons), nn.ReLU()]
        for _ in range(num_hidden_layers):
            layers += [nn.Linear(num_neurons, num_neurons), nn.LayerNorm(num_neurons), nn.ReLU()]
        layers.append(nn.Linear(num_neurons, out))
        self.model = nn.Sequential(*layers)
        _he_init(self.model)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.model(x).view(x.shape[0], self.num_groups, self.num_biquads, self.num_params)


class ResidualBlock(nn.Module):
    """reference dnn.py:267-281."""

    def __init__(self, num_neurons: int):
        super().__init__()
        self.linear = nn.Linear(num_neurons, num_neurons)
        self.norm = nn.LayerNorm(num_neurons)
        self.activation = nn.ReLU()

    def forward(self, x):
        return self.activation(self.norm(self.linear(x))) + x


class MLP_SkipConnections(nn.Module):
    """reference dnn.py:284-328 (input_layer / hidden_layers.{i}.{linear,norm} / output_layer)."""

    def __init__(self, num_pos_features: int, num_hidden_layers: int, num_neurons: int,
                 num_groups: int, num_biquads_in_cascade: int, num_params: int):
        super().__init__()
        self.num_biquads = num_biquads_in_cascade
        self.num_groups = num_groups
        self.num_params = num_params
        out = num_groups * num_params * num_biquads_in_cascade
        self.input_layer = nn.Sequential(nn.Linear(num_pos_features, num_neurons),
                                         nn.LayerNorm(num_neurons), nn.ReLU())
        self.hidden_layers = nn.ModuleList([ResidualBlock(num_neurons) for _ in range(num_hidden_layers)])
        self.output_layer = nn.Linear(num_neurons, out)
        _he_init(self)

    def forward(self, x):
        b = x.shape[0]
        x = self.input_layer(x)
        for layer in self.hidden_layers:
            x = layer(x)
        return self.output_layer(x).view(b, self.num_groups, self.num_biquads, self.num_params)
