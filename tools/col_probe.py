import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffgfdn_amd import hip_ops as ops
dev = torch.device('cuda', 0)
torch.manual_seed(0)
H = torch.randn(32, 65537, dtype=torch.complex64, device=dev)
for _ in range(6):
    x = ops.irfft_odd_fwd(H, 65537)
torch.cuda.synchronize()
