"""Hot-loop timing of the small kernels of the step (diagnostic): back-to-back launches, HIP events."""
import os, sys, time
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffgfdn_amd import hip_ops as ops
dev = torch.device('cuda', 0)
nb, Bper, G, F, H, nh = 7, 32, 4, 20, 16, 5
B = nb * Bper
g = torch.Generator().manual_seed(0)
P = ops._lib.load().gfdn_mlp_param_count(F, H, nh, G)
w = (0.3 * torch.randn(nb, P, generator=g)).to(dev)
R = 838
pos = torch.rand(nb * R, 3, generator=g, dtype=torch.float64).to(dev)
rows = torch.stack([q * R + torch.randperm(R, generator=g)[:Bper] for q in range(nb)]).reshape(-1).to(dev)
fpi = (torch.exp(torch.linspace(0, np.log(32.0), F)) * np.pi).to(dev)
big = torch.empty(512 * 1024 * 1024 // 4, device=dev)

def timeit(fn, n=200, flush=False):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    if not flush:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    tot = 0.0
    for _ in range(20):
        big.add_(1.0)                      # 1 GB of traffic: caches and TLBs cold
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return tot / 20

gains, xhat, rstd = ops.mlp_gains_fwd(pos, fpi, w, H, nh, G, -1.0, 1.0, rows, nbands=nb)
gg = torch.randn(B, G, generator=g).to(dev)
out = torch.empty(nb, P, device=dev)
M = (0.4 * torch.randn(nb * G, 4, 4, generator=g)).to(dev)
b = torch.randn(nb * G * 4, generator=g).to(dev); c = torch.randn(nb * G * 4, generator=g).to(dev)
ig = (1.0 + 0.01 * torch.rand(nb * G * 4, generator=g)).to(dev)
for name, fn in [('mlp_fwd', lambda: ops.mlp_gains_fwd(pos, fpi, w, H, nh, G, -1.0, 1.0, rows, nbands=nb)),
                 ('mlp_bwd', lambda: ops.mlp_gains_bwd(pos, fpi, w, H, nh, G, -1.0, 1.0, gains, xhat, rstd, gg, rows, nbands=nb, out=out)),
                 ('ortho_coefs', lambda: ops.tf_ortho_coefs(M, ig, b, c)),
                 ('empty-ish (fill 64 floats)', lambda: out[0, :64].zero_())]:
    print(f"{name:28s} hot {timeit(fn):7.2f} us   cold {timeit(fn, flush=True):7.2f} us")
