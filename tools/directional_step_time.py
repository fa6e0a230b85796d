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
if len(sys.argv) > 1 and sys.argv[1] == 'graph':
    pass
for it in range(0 if (len(sys.argv) > 1 and sys.argv[1] == 'graph') else 8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tot, _ = tr.train_step(batch)
    torch.cuda.synchronize(); print(f'directional eager step {1e3*(time.perf_counter()-t0):.2f} ms loss {float(tot):.4f}', flush=True)

# the same step replayed from a HIP graph (static batch tensors; capturable flat Adam).  Warm-up and capture run on ONE
# stream: autograd pins every parameter's AccumulateGrad node to the stream of its first backward, and a capture that
# has to hop to that stream and back dies in hipStreamEndCapture.
if len(sys.argv) > 1 and sys.argv[1] == 'graph':
    net = DiffDirectionalFDNVarReceiverPos(fs, G, delays, dev, fl, of, ambi_order=order, common_decay_times=T60,
                                           use_colorless_loss=True, analysis_matrix=A).to(dev)
    tr2 = DirectionalFDNVarReceiverPosTrainer(net, tc, capturable=True)
    cs = torch.cuda.Stream()
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        for _ in range(3):
            tr2.train_step(batch)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cs):
        tot, parts = tr2.train_step(batch)
    torch.cuda.synchronize()
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize(); print(f'directional graph step {1e3*(time.perf_counter()-t0)/20:.3f} ms loss {float(tot):.4f}', flush=True)
    sys.exit(0)
