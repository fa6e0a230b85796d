"""Isolated timings of the block-transfer-function kernels at the bench size (7 bands x 4 groups x 4 lines, 32 receivers
per band, K = 65 537 / 32 769 bins).  usage: python tools/tf_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffgfdn_amd import hip_ops as ops
dev = 'cuda'
nb, G, n, B, K = 7, 4, 4, 32, 65537
Ku = (K + 1) // 2
nblk = nb * G
g = torch.Generator().manual_seed(0)
z = torch.polar(torch.ones(K, dtype=torch.float64), np.pi * torch.arange(K, dtype=torch.float64) / (K - 1)).to(dev)
turns, _ = ops.zprep(z)
tu = turns[:Ku].contiguous()
X = torch.triu((2 * torch.rand(nblk, n, n, generator=g) - 1) / 2, 1)
Q = torch.linalg.matrix_exp(X - X.transpose(1, 2))
A = (Q @ Q).to(dev)
M = ((2 * torch.rand(nblk, n, n, generator=g) - 1) / 2).to(dev)
b = ((2 * torch.randn(nblk * n, generator=g) - 1) / 16).to(dev)
c = ((2 * torch.randn(nblk * n, generator=g) - 1) / 16).to(dev)
delays = torch.tensor(np.random.RandomState(0).choice(np.arange(641, 1601), nblk * n), dtype=torch.float32, device=dev)
ig = (1.0 / 10 ** (-3 * delays / (32000.0 * 1.0))).to(dev)
coef = ops.tf_coefs(A, b, c, ig)
coef_s = ops.tf_coefs(M, b, c)
R = 838
direct = torch.view_as_complex(torch.randn(nb * R, Ku, 2, device=dev))
rows = torch.cat([q * R + torch.randperm(R, generator=g)[:B] for q in range(nb)]).to(dev)
rows_seq = torch.cat([q * R + torch.arange(B) for q in range(nb)]).to(dev)
filt = torch.view_as_complex(torch.randn(nb, Ku, 2, device=dev))
rgain = torch.rand(nb * B, G, device=dev)
gH = torch.view_as_complex(torch.randn(nb * B, Ku, 2, device=dev))
scale = torch.rand(nblk, device=dev) + 0.5

def bench(name, fn, nbytes=None, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / it
    extra = f"  {nbytes / us / 1e6:.2f} TB/s" if nbytes else ""
    print(f"{name:42s} {us:8.1f} us{extra}", flush=True)

H = torch.empty(nb * B, Ku, dtype=torch.complex64, device=dev)
hb = nb * B * Ku * 8
bench("copy H->H (torch)", lambda: H.copy_(gH), 2 * hb)
bench("tf_compose_fwd random rows", lambda: ops.tf_compose_fwd(tu, None, coef, delays, n, rgain, scale, direct, filt, rows, nb, out=H), 2 * hb)
bench("tf_compose_fwd sequential rows", lambda: ops.tf_compose_fwd(tu, None, coef, delays, n, rgain, scale, direct, filt, rows_seq, nb, out=H), 2 * hb)
bench("tf_compose_fwd no direct", lambda: ops.tf_compose_fwd(tu, None, coef, delays, n, rgain, scale, None, filt, None, nb, out=H), hb)
_, Ts = ops.tf_compose_fwd(tu, None, coef, delays, n, rgain, scale, direct, filt, rows, nb, save_T=True)
bench("tf_gain_grad", lambda: ops.tf_gain_grad(Ts, gH, G, filt, nb), hb)
bench("tf_compose_bwd (records)", lambda: ops.tf_compose_bwd(tu, None, coef, delays, n, rgain, gH, Ts, filt, nb), hb)
bench("tf_eval main (Ku)", lambda: ops.tf_eval(tu, None, coef, delays, n))
bench("tf_eval sub (K)", lambda: ops.tf_eval(turns, None, coef_s, delays, n))
bb, cc = b.clone(), c.clone()
bench("tf_energy (K)", lambda: ops.tf_energy(turns, None, coef_s, delays, n))
bench("tf_energy (K) runs", lambda: ops.tf_energy(turns, None, coef_s, delays, n, dturn=0.5 / (K - 1)))
bench("tf_colorless (K)", lambda: ops.tf_colorless(turns, None, coef_s, delays, n, scale, True, 1.0))
bench("tf_colorless (K) runs", lambda: ops.tf_colorless(turns, None, coef_s, delays, n, scale, True, 1.0, dturn=0.5 / (K - 1)))
bench("tf_coefs", lambda: ops.tf_coefs(A, b, c, ig))
grec = torch.randn(nblk, 32, device=dev)
bench("tf_coefs_bwd 2 sets", lambda: ops.tf_coefs_bwd(A, ig, grec, b, c, A1=M, grec1=grec))
