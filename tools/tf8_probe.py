"""Times the passes of csrc/blocktf8.hip alone on the chip at the N = 32 bench shape (28 blocks of 8 lines, K = 65 537 /
32 769 bins, 32 receivers per band).      usage: python tools/tf8_probe.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffgfdn_amd import _lib
if '--lib' in sys.argv:
    _lib.LIB_PATH = sys.argv[sys.argv.index('--lib') + 1]
from diffgfdn_amd import hip_ops as ops

dev = 'cuda'
nb, G, n, B, K = 7, 4, 8, 32, 65537
Ku = (K + 1) // 2
nblk = nb * G
g = torch.Generator().manual_seed(0)
M = (0.4 * torch.randn(nblk, n, n, generator=g)).to(dev)
Q, QQ = ops.ortho_fwd(M, True, True)
delays = torch.tensor(np.sort(np.random.RandomState(0).choice(np.arange(571, 1214), nblk * n, replace=True)).astype(np.float32)).to(dev)
ig = (1.0 / 10 ** (-3 * delays / (32000.0 * 0.9))).to(dev)
b = ((2 * torch.randn(nblk * n, generator=g) - 1) / 32).to(dev)
c = ((2 * torch.randn(nblk * n, generator=g) - 1) / 32).to(dev)
z = torch.polar(torch.ones(K, dtype=torch.float64), np.pi * torch.arange(K, dtype=torch.float64) / (K - 1)).to(dev)
turns, _ = ops.zprep(z)
turns_u = turns[:Ku].contiguous()
coef, coef_sub = ops.tf8_coefs(QQ, ig, b, c, A1=M)
scale = torch.ones(nblk, device=dev)
rgain = (2 * torch.rand(nb * B, G, generator=g) - 1).to(dev)
gH = torch.randn(nb * B, Ku, dtype=torch.complex64, device=dev)
filt = torch.randn(nb, Ku, dtype=torch.complex64, device=dev)


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / it


bb, cc = b.clone(), c.clone()
print(f"coefs (2 sets)        {timed(lambda: ops.tf8_coefs(QQ, ig, b, c, A1=M)):8.1f} us")
print(f"energy   K = {K}  {timed(lambda: ops.tf8_energy(turns, coef_sub, delays, n, bb.copy_(b), cc.copy_(c), dturn=0.5 / (K - 1))):8.1f} us")
print(f"tsave    K = {Ku}  {timed(lambda: ops.tf8_tsave(turns_u, coef, delays, n, c, scale, nb, G)):8.1f} us")
print(f"colorless K = {K} {timed(lambda: ops.tf8_colorless(turns, coef_sub, delays, n, c, scale, True, 1.0, dturn=0.5 / (K - 1))):8.1f} us")
print(f"bwd      K = {Ku}  {timed(lambda: ops.tf8_compose_bwd(turns_u, coef, delays, n, c, scale, rgain, gH, filt, nb)):8.1f} us")
part = ops.tf8_compose_bwd(turns_u, coef, delays, n, c, scale, rgain, gH, filt, nb)
part2, _ = ops.tf8_colorless(turns, coef_sub, delays, n, c, scale, True, 1.0)
print(f"param grads           {timed(lambda: ops.tf8_param_grads(QQ, ig, part, b, c, M, A1=M, part1=part2, Q=Q)):8.1f} us")
