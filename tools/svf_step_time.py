"""Step time of the full-band grid model with SVF output filters at full size (K = 65 537, 32 receivers):
python tools/svf_step_time.py [eager|graph]   (1.2 ms graph replay, 4.3 ms eager)"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from diffgfdn_amd.config import (CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig, TrainerConfig)
from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset, split_dataset
from diffgfdn_amd.model import DiffGFDNVarReceiverPos
from diffgfdn_amd.synthetic import synthetic_room
from diffgfdn_amd.trainer import VarReceiverPosTrainer
dev = torch.device('cuda', 0)
G, FS, NFFT, BATCH = bench.G, bench.FS, bench.NFFT, bench.BATCH
room = synthetic_room(128, G, FS, 64000, seed=0)
ds = RoomDataset(G, FS, room['source_position'], room['receiver_position'], room['rirs'], room['common_decay_times'], nfft=NFFT, device=dev)
data = MultiRIRDataset(dev, ds)
torch.manual_seed(1)
delays = [1009, 1153, 1289, 1427, 1531, 1663, 1789, 1907, 2011, 2141, 2269, 2393, 2521, 2647, 2767, 2887]
fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
of = OutputFilterConfig(use_svfs=True, num_hidden_layers=5, num_neurons_per_layer=16, num_fourier_features=20, compress_pole_factor=0.98)
net = DiffGFDNVarReceiverPos(FS, G, delays, dev, fl, of, use_absorption_filters=False, common_decay_times=room['common_decay_times'], use_colorless_loss=True).to(dev)
tc = TrainerConfig(batch_size=BATCH, num_freq_bins=NFFT, max_epochs=1, lr=1e-3, io_lr=1e-2, use_edc_mask=True, use_colorless_loss=True,
                   edc_loss_weight=10, sparsity_loss_weight=2, use_asym_spectral_loss=True, device='cuda', train_dir='/tmp/gfdn_svf/t', ir_dir='/tmp/gfdn_svf/i')
tr = VarReceiverPosTrainer(net, tc, capturable=True)
mode = sys.argv[1] if len(sys.argv) > 1 else 'eager'
idx = list(range(BATCH))
if mode == 'eager':
    batch = data.collate(idx)
    tr.normalize(batch)
    for it in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tot, _ = tr.train_step(batch)
        torch.cuda.synchronize(); print(f'eager step {1e3*(time.perf_counter()-t0):.2f} ms loss {float(tot):.3f}', flush=True)
else:
    batch = data.collate(idx)
    tr.normalize(batch)
    step = tr.graphed(data, BATCH)
    for it in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = step(idx)
        torch.cuda.synchronize(); print(f'graph step {1e3*(time.perf_counter()-t0):.2f} ms loss {float(out["_total"]):.3f}', flush=True)
