"""Times gfdn_edc_lin_one (and the light gamma sweep) alone on the chip at the bench shape, for library variants."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffgfdn_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != 'default':
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', '_probe', sys.argv[1])
from diffgfdn_amd import hip_ops as ops
DEV = 'cuda'
gen = torch.Generator(device='cpu').manual_seed(1)
nb, B, G, n, R, start, L = 7, 32, 4, 65537, 838, 640, 47360
items = nb * B
decay = torch.exp(-torch.arange(n) / 9000.0)
xd = (torch.randn(nb * R, n, generator=gen) * decay).to(DEV)
tau = (torch.randn(nb * G // 2, n, 2, generator=gen) * decay[None, :, None]).to(DEV)
rgain = torch.randn(items, G, generator=gen).to(DEV)
rows = torch.tensor([q * R + int(i) for q in range(nb) for i in torch.randperm(R, generator=gen)[:B]], device=DEV)
T_db = (-30 * torch.rand(nb * R, L, generator=gen)).to(DEV)
mw = ((torch.rand(L, generator=gen) > 0.5).float() / L).to(DEV)
parts = torch.zeros(items * G, 34, device=DEV)
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
t1 = timeit(lambda: ops.edc_lin_one(xd, rows, tau, rgain, nb, n, start, L, T_db, mw, 1.0, 10.0, True, trows=rows, dots=parts, col=0))
li, gx = ops.edc_lin_one(xd, rows, tau, rgain, nb, n, start, L, T_db, mw, 1.0, 10.0, True, trows=rows, dots=parts, col=0)
t2 = timeit(lambda: ops.lin_gamma_win(gx, rgain, nb, n, start, L))
t3 = timeit(lambda: ops.edc_loss_pairs_lin(xd, rows, tau, rgain, nb, n, start, L, T_db, mw, 1.0, 10.0, True, trows=rows, fill_outside=False))
print(sys.argv[1:], "edc_lin_one us %.1f  gamma_win us %.1f  three-launch form us %.1f (incl. alloc)" % (t1, t2, t3))
