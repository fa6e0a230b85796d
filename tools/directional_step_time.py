"""Eager step time of the directional model (4 groups x 9 SH channels, 16 directions, 32 receivers, K = 65 537):
python tools/directional_step_time.py   (4.1 ms)"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig, TrainerConfig
from diffgfdn_amd.model import DiffDirectionalFDNVarReceiverPos
from diffgfdn_amd.trainer import DirectionalFDNVarReceiverPosTrainer
from diffgfdn_amd.losses import directional_edc_loss
dev = 'cuda'
fs, G, order, B, nfft, J = 32000.0, 4, 2, 32, 131072, 16
L = (order + 1) ** 2
K = nfft // 2 + 1
rng = np.random.RandomState(0)
delays = sorted(rng.choice(np.arange(900, 3000), G * L, replace=False).tolist())
T60 = np.array([[0.5, 0.8, 1.1, 1.4]])
fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
of = OutputFilterConfig(use_svfs=False, num_hidden_layers=5, num_neurons_per_layer=16, num_fourier_features=20)
A = rng.randn(J, L).astype(np.float32)
net = DiffDirectionalFDNVarReceiverPos(fs, G, delays, dev, fl, of, ambi_order=order, common_decay_times=T60,
                                       use_colorless_loss=True, analysis_matrix=A).to(dev)
tc = TrainerConfig(use_colorless_loss=False, edc_loss_weight=1.0, train_dir='/tmp/gfdn_t', ir_dir='/tmp/gfdn_a', device='cuda',
                   lr=1e-3, io_lr=1e-2)
tr = DirectionalFDNVarReceiverPosTrainer(net, tc)
z = torch.exp(1j * np.pi * torch.arange(K, dtype=torch.float64) / (K - 1)).to(dev)
batch = {'z_values': z, 'listener_position': torch.rand(B, 3, device=dev, dtype=torch.float64) * 5,
         'norm_listener_position': torch.rand(B, 3, device=dev, dtype=torch.float64),
         'source_position': torch.zeros(B, 3, device=dev, dtype=torch.float64),
         'target_early_response': torch.randn(B, K, dtype=torch.complex128, device=dev) * 0.01,
         'target_common_slope_amps': torch.rand(B, J, G, dtype=torch.float64, device=dev)}
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tot, _ = tr.train_step(batch)
    torch.cuda.synchronize(); print(f'directional eager step {1e3*(time.perf_counter()-t0):.2f} ms loss {float(tot):.4f}', flush=True)
