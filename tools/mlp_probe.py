"""Times the gain network launches of a bank with per-band layer sizes (csrc/mlp.hip k_mlp_bands_*) for subsets of the
reference's recipe, alone on the chip.   usage: python tools/mlp_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from diffgfdn_amd import hip_ops as ops
DEV = 'cuda'
g = torch.Generator(device='cpu').manual_seed(1)
R, Bper, G, F = 838, 32, 3, 20


def run(sizes, tag):
    nb = len(sizes)
    H, nh = [m[0] for m in sizes], [m[1] for m in sizes]
    counts = ops.mlp_bands_param_counts(H, nh, F, G)
    w = torch.cat([0.3 * torch.randn(c, generator=g) / np.sqrt(h) for c, h in zip(counts, H)]).to(DEV)
    pos = torch.rand(nb * R, 3, generator=g, dtype=torch.float64).to(DEV)
    fpi = (torch.exp(torch.linspace(0, np.log(32.0), F)) * np.pi).to(torch.float32).to(DEV)
    rows = torch.tensor([q * R + int(i) for q in range(nb) for i in torch.randperm(R, generator=g)[:Bper]], device=DEV)
    parts = torch.randn(nb * Bper * G, 10, generator=g).to(DEV)
    cs = (torch.rand(nb * G, generator=g) + 0.5).to(DEV)
    hetero = len(set(sizes)) > 1
    Ha, nha = (H, nh) if hetero else (H[0], nh[0])
    gains, xhat, rstd = ops.mlp_gains_fwd(pos, fpi, w, Ha, nha, G, -1.0, 1.0, rows, nbands=nb)

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    big = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=DEV)

    def cold(fn, reps=10):
        tot = 0.0
        for _ in range(reps):
            big.add_(1.0)                      # (512 MB through the caches: the next launch starts cold)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            tot += a.elapsed_time(b)
        return tot / reps * 1e3
    tfc = cold(lambda: ops.mlp_gains_fwd(pos, fpi, w, Ha, nha, G, -1.0, 1.0, rows, nbands=nb, out=(gains, xhat, rstd)))
    print(f"{tag:28s} forward from cold caches {tfc:7.1f} us (host-paired: + ~5 us of launch)")
    tf = timeit(lambda: ops.mlp_gains_fwd(pos, fpi, w, Ha, nha, G, -1.0, 1.0, rows, nbands=nb, out=(gains, xhat, rstd)))
    tb = timeit(lambda: ops.mlp_gains_bwd(pos, fpi, w, Ha, nha, G, -1.0, 1.0, gains, xhat, rstd, None, rows, nbands=nb,
                                          ggains_parts=parts, colscale=cs))
    print(f"{tag:28s} forward {tf:7.1f} us   backward + reduce {tb:7.1f} us")


run([(8, 1), (16, 1), (16, 5), (16, 5), (16, 5), (128, 3), (128, 3), (128, 3)], "reference recipe (8 bands)")
run([(8, 1), (16, 1), (16, 5), (16, 5), (16, 5)], "the five small bands")
run([(128, 3), (128, 3), (8, 1)], "two 3 x 128 bands + 1 x 8")
run([(16, 5)] * 7, "7 x (5 x 16), one MlpDims")
run([(8, 1), (16, 1)], "1 x 8 + 1 x 16")
