"""Phase stamps of k_edc_lin_one / k_edr_lin_wave alone on the chip (probe libs lib_e1t.so / lib_edwt.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from diffgfdn_amd import _lib
which = sys.argv[1]
_lib.LIB_PATH = os.path.join(ROOT, 'tools', '_probe', sys.argv[2])
from diffgfdn_amd import hip_ops as ops
DEV = 'cuda'
gen = torch.Generator(device='cpu').manual_seed(1)
nb, B, G, n, R, start, L = 7, 32, 4, 65537, 838, 640, 47360
items = nb * B
rgain = torch.randn(items, G, generator=gen).to(DEV)
rows = torch.tensor([q * R + int(i) for q in range(nb) for i in torch.randperm(R, generator=gen)[:B]], device=DEV)
lib = _lib.load()
def pr(name, d):
    print("%-14s median %6.2f  mean %6.2f  10%% %6.2f  90%% %6.2f  max %6.2f" % ((name, np.median(d), d.mean()) + tuple(np.percentile(d, [10, 90, 100]))))
if which == 'edc':
    decay = torch.exp(-torch.arange(n) / 9000.0)
    xd = (torch.randn(nb * R, n, generator=gen) * decay).to(DEV)
    tau = (torch.randn(nb * G // 2, n, 2, generator=gen) * decay[None, :, None]).to(DEV)
    T_db = (-30 * torch.rand(nb * R, L, generator=gen)).to(DEV)
    mw = ((torch.rand(L, generator=gen) > 0.5).float() / L).to(DEV)
    parts = torch.zeros(items * G, 34, device=DEV)
    for _ in range(6):
        ops.edc_lin_one(xd, rows, tau, rgain, nb, n, start, L, T_db, mw, 1.0, 10.0, True, trows=rows, dots=parts, col=0)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 8))()
    lib.gfdn_probe_edc_one_times.restype = ctypes.c_int
    rc = lib.gfdn_probe_edc_one_times(buf, 1024 * 8)
    t = np.array(buf, dtype=np.int64).reshape(1024, 8)[:items].astype(np.float64) / 100.0
    t0 = t[:, 0].min()
    print("rc", rc); pr("start", t[:, 0] - t0)
    for i, nm in enumerate(['loads+compose+scan', 'tables', 'dB stage+rev scan', 'tables 2', 'dL/dx + dots']):
        pr(nm, t[:, i + 1] - t[:, i])
    pr("lifetime", t[:, 5] - t[:, 0]); pr("end", t[:, 5] - t0)
else:
    nfr, nf = 32, 2049
    Sd = torch.randn(nb * R, nfr, nf, 2, generator=gen).to(DEV)
    Sd = torch.view_as_complex(Sd)
    Stau = torch.view_as_complex(torch.randn(nb * G, nfr, nf, 2, generator=gen).to(DEV))
    T = (-30 * torch.rand(nb * R, nfr, nf, generator=gen)).to(DEV)
    sa = (T.abs().sum((1, 2)))
    parts = torch.zeros(items * G, 1 + lib.gfdn_edr_lin_band_parts(nf), device=DEV)
    for _ in range(6):
        ops.edr_lin_loss_gsum(Sd, rows, Stau, rgain, nb, T, sa, 1.0, dots=parts, col0=1, tiled=True, nsplit=2)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 4))()
    lib.gfdn_probe_edr_wave_times.restype = ctypes.c_int
    rc = lib.gfdn_probe_edr_wave_times(buf, 1024 * 4)
    t = np.array(buf, dtype=np.int64).reshape(1024, 4).astype(np.float64) / 100.0
    t0 = t[:, 0].min()
    print("rc", rc); pr("start", t[:, 0] - t0)
    t = t[t[:, 0] > 0]                     # (910 workgroups of 1024 slots)
    t0 = t[:, 0].min()
    pr("start", t[:, 0] - t0)
    pr("receivers 0, 1", t[:, 1] - t[:, 0]); pr("receivers 2..15", t[:, 2] - t[:, 1]); pr("per receiver", (t[:, 2] - t[:, 1]) / 14)
    pr("end of loop", t[:, 2] - t0)
