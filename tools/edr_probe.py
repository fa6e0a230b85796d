"""Times gfdn_edr_lin_loss_gsum alone on the chip at the bench shape, for library variants."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffgfdn_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != 'default':
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', '_probe', sys.argv[1])
from diffgfdn_amd import hip_ops as ops
DEV = 'cuda'
gen = torch.Generator(device='cpu').manual_seed(1)
nb, B, G, R = 7, 32, 4, 838
items = nb * B
nfr, nf = 32, 2049
lib = _lib.load()
rgain = torch.randn(items, G, generator=gen).to(DEV)
rows = torch.tensor([q * R + int(i) for q in range(nb) for i in torch.randperm(R, generator=gen)[:B]], device=DEV)
Sd = torch.view_as_complex(torch.randn(nb * R, nfr, nf, 2, generator=gen).to(DEV))
Stau = torch.view_as_complex(torch.randn(nb * G, nfr, nf, 2, generator=gen).to(DEV))
T = (-30 * torch.rand(nb * R, nfr, nf, generator=gen)).to(DEV)
sa = T.abs().sum((1, 2))
parts = torch.zeros(items * G, 1 + lib.gfdn_edr_lin_band_parts(nf), device=DEV)
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
t1 = timeit(lambda: ops.edr_lin_loss_gsum(Sd, rows, Stau, rgain, nb, T, sa, 1.0, dots=parts, col0=1, tiled=True, nsplit=2))
part, Gs = ops.edr_lin_loss_gsum(Sd, rows, Stau, rgain, nb, T, sa, 1.0, dots=parts, col0=1, tiled=True, nsplit=2)
print(sys.argv[1:], "edr_lin_loss_gsum us %.1f (incl. alloc)  checksum %.6e %.6e %.6e" % (t1, float(part.double().sum()), float(Gs.abs().double().sum()), float(parts.double().sum())))
