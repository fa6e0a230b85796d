"""Colorless (lossless-prototype) FDN on the MI355X hot path.

Interface mirror of the reference's src/diff_gfdn/colorless_fdn/: ``ColorlessFDN`` (model.py:12-111: one
group, dense orthogonal feedback matrix Q = expm(skew(W)) -- CouplingMatrixType.RANDOM -- shared nominal
T60, learnable input / output gains), ``ColorlessFDNTrainer`` (trainer.py:17-160: spectral (a)MSE against a
flat unit response + alpha x sparsity of Q, energy normalisation of b, c at construction, Adam + StepLR, early
stopping), ``ColorlessFDNDataset`` (dataloader.py:6-34: frequency points on the upper unit circle, unit
labels), ``ColorlessFDNResults`` (utils.py:11-15).  Its optimised (b, c, Q) initialise the groups of a
DiffGFDN (``colorless_fdn_params`` of the models).

Same kernels as the grid model: the per-bin solve y = (D Gamma^-1 - Q)^-1 b on one dense N x N block
(``gfdn_solve_fwd/bwd``, N <= 32), the output stage as the sum c^T y, the spectral statistics kernel.
"""
import os
import time
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch
from torch import nn

from .colorless_losses import amse_loss, mse_loss, sparsity_loss
from .config import CouplingMatrixType, TrainerConfig
from .feedback_loop import FeedbackLoop, decay_times_to_gain_per_sample
from .functional import OutputStage


@dataclass
class ColorlessFDNResults:
    """reference colorless_fdn/utils.py:11-15"""
    opt_input_gains: torch.Tensor
    opt_output_gains: torch.Tensor
    opt_feedback_matrix: torch.Tensor


def get_frequency_samples(num: int, device=None) -> torch.Tensor:
    """z = exp(i pi linspace(0, 1, num))   (reference utils.py:128-141)."""
    angle = torch.linspace(0, 1, steps=num, device=device)
    return torch.polar(torch.ones(num, device=device), angle * np.pi)


class ColorlessFDNDataset(torch.utils.data.Dataset):
    """Frequency points exp(i pi k / num), k < num, with unit labels (reference dataloader.py:6-34)."""

    def __init__(self, num_freq_samples: int, device):
        angle = torch.arange(0, 1, 1 / num_freq_samples)
        self.labels = torch.ones(num_freq_samples).to(device)
        self.input = torch.polar(torch.ones(num_freq_samples), angle * np.pi).to(device)

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, index):
        return self.input[index], self.labels[index]


class ColorlessFDN(nn.Module):
    """reference colorless_fdn/model.py:12-111."""

    def __init__(self, sample_rate: int, delays: List[int], device: torch.device, nominal_t60: float = 10.0):
        super().__init__()
        self.sample_rate = sample_rate
        self.device = device
        self.num_delay_lines = len(delays)
        N = self.num_delay_lines
        if N > 32:
            raise NotImplementedError("the per-bin solver takes dense blocks up to 32 x 32")
        # draws in the reference's order: input gains, output gains, then the feedback loop's matrix
        self.input_gains = nn.Parameter((2 * torch.randn(N, 1) - 1) / N)
        self.output_gains = nn.Parameter((2 * torch.randn(N, 1) - 1) / N)
        self.gain_per_sample = torch.tensor(decay_times_to_gain_per_sample(nominal_t60, delays, sample_rate))
        self.feedback_loop = FeedbackLoop(sample_rate, num_groups=1, num_delay_lines_per_group=N,
                                          delays=torch.tensor(delays, dtype=torch.float32),
                                          gains=self.gain_per_sample, use_absorption_filters=False,
                                          coupling_matrix_type=CouplingMatrixType.RANDOM)
        self._one = None

    @property
    def delays(self) -> torch.Tensor:
        return self.feedback_loop.delays

    def forward(self, z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(H (K,), H_per_del (N, K)):  H = c^T P b,  H_per_del[n] = c_n (P b)_n   (reference :63-92)."""
        z = z.to(torch.complex128)
        self.feedback_loop.new_forward()
        Y = self.feedback_loop.resolvent_apply(z, self.input_gains)             # (K, N): y = P b
        if self._one is None or self._one.device != Y.device:
            self._one = torch.ones((1, 1), dtype=torch.float32, device=Y.device)
        H = OutputStage.apply(Y, self.output_gains.reshape(-1), self._one, self.num_delay_lines, None, None)
        H_per_del = (Y * self.output_gains.reshape(1, -1)).T
        return H.reshape(-1), H_per_del

    @torch.no_grad()
    def get_param_dict(self) -> Dict:
        return {'delays': self.delays.squeeze().cpu().numpy(),
                'gains_per_sample': self.gain_per_sample.squeeze().cpu().numpy(),
                'input_gains': self.input_gains.squeeze().cpu().numpy(),
                'output_gains': self.output_gains.squeeze().cpu().numpy(),
                'feedback_matrix': self.feedback_loop.random_feedback_matrix.squeeze().cpu().numpy()}

    @torch.no_grad()
    def results(self) -> ColorlessFDNResults:
        """What the DiffGFDN models take as ``colorless_fdn_params`` (reference utils.py:18-60)."""
        return ColorlessFDNResults(self.input_gains.detach().squeeze().cpu(), self.output_gains.detach().squeeze().cpu(),
                                   self.feedback_loop.random_feedback_matrix.detach().cpu().numpy())


class ColorlessFDNTrainer:
    """reference colorless_fdn/trainer.py:17-160 (``alpha``, ``lr``, ``max_epochs``, ``batch_size`` are the
    fields of its ColorlessFDNConfig)."""

    def __init__(self, net: ColorlessFDN, trainer_config: TrainerConfig, alpha: float = 1.0, lr: float = 0.01,
                 max_epochs: int = 20, batch_size: int = 2000):
        self.net = net
        self.device = trainer_config.device
        self.max_epochs = max_epochs
        self.patience = 5
        self.early_stop = 0
        self.alpha = alpha
        self.train_dir = trainer_config.train_dir + "colorless-fdn/"
        self.optimizer = torch.optim.Adam(self.net.parameters(), lr=lr)
        spec = amse_loss() if trainer_config.use_asym_spectral_loss else mse_loss()
        self.criterion = [spec, sparsity_loss()]
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=10, gamma=0.1)
        dev = next(net.parameters()).device
        self.z = get_frequency_samples(int(net.sample_rate * 2), device=dev)
        self.z_batch = get_frequency_samples(batch_size, device=dev)
        self.normalize()

    def _loss(self, inputs, labels, with_lines: bool):
        H, H_per_del = self.net(inputs)
        fl = self.net.feedback_loop
        loss = self.criterion[0](H, labels) + self.alpha * self.criterion[1](fl.ortho_param(fl.random_feedback_matrix))
        if with_lines:
            loss = loss + self.criterion[0](H_per_del, torch.ones_like(H_per_del))
        return loss

    def train_step(self, data) -> float:
        inputs, labels = data
        self.optimizer.zero_grad()
        loss = self._loss(inputs, labels, False)
        loss.backward()
        self.optimizer.step()
        return loss.item()

    @torch.no_grad()
    def valid_step(self, data) -> float:
        inputs, labels = data
        return self._loss(inputs, labels, True).item()

    @torch.no_grad()
    def normalize(self):
        """b, c /= (mean_k |H(z_k)|^2)^(1/4) over the batch grid (reference :133-143)."""
        H, _ = self.net(self.z_batch)
        energy = torch.sum(torch.abs(H) ** 2) / H.shape[0]
        for name, prm in self.net.named_parameters():
            if name in ('input_gains', 'output_gains'):
                prm.data.copy_(prm.data / torch.pow(energy, 1 / 4))

    def train(self, train_dataset, valid_dataset):
        self.train_loss, self.valid_loss = [], []
        st = time.time()
        for epoch in range(self.max_epochs):
            tot = sum(self.train_step(d) for d in train_dataset)
            self.scheduler.step()
            self.train_loss.append(tot / max(len(train_dataset), 1))
            tot = sum(self.valid_step(d) for d in valid_dataset)
            self.valid_loss.append(tot / max(len(valid_dataset), 1))
            self.save_model(epoch)
            if epoch >= 1:
                self.early_stop = self.early_stop + 1 if abs(self.valid_loss[-2] - self.valid_loss[-1]) <= 1e-4 else 0
            if self.early_stop == self.patience:
                break
        self.train_time = time.time() - st

    @torch.no_grad()
    def save_ir(self, directory: str, filename: str = 'colorless_fdn_ir.wav', norm: bool = False):
        """Impulse response of the prototype on the trainer's grid as a 32-bit float stereo wav (reference :146-160).
        The inverse transform has length n = 2 (K - 1): the power-of-two kernel (gfdn_irfft_pow2_fwd) when n is one;
        otherwise (the reference's default grid has 2 fs points) the response is brought to the host, where the wav is
        written anyway, and transformed there -- no FFT library runs on the device on any path of this package."""
        from scipy.io import wavfile
        from . import hip_ops as ops
        H, _ = self.net(self.z)
        n = 2 * (H.shape[-1] - 1)
        if n >= 8 and n & (n - 1) == 0 and H.is_cuda:
            h = ops.irfft_pow2_fwd(H.reshape(1, -1).to(torch.complex64), n).reshape(-1)
        else:
            h = torch.from_numpy(np.fft.irfft(H.detach().cpu().numpy().astype(np.complex128), n=n)).to(H.device)
        if norm:
            h = h / torch.max(torch.abs(h))
        os.makedirs(directory, exist_ok=True)
        hc = h.to(torch.float32).cpu().numpy()
        wavfile.write(os.path.join(directory, filename), int(self.net.sample_rate), np.stack((hc, hc), axis=1))
        return h

    def save_model(self, e: int):
        d = os.path.join(self.train_dir, 'checkpoints')
        os.makedirs(d, exist_ok=True)
        torch.save(self.net.state_dict(), os.path.join(d, 'model_e' + str(e) + '.pt'))
