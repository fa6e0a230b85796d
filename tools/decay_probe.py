"""Times the fused decay kernel (csrc/decay.hip) alone on the chip against the unfused kernels it replaces, at the
bench shape (224 items, n = 65 537).      usage: python tools/decay_probe.py [items]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffgfdn_amd import hip_ops as ops

dev = 'cuda'
items = int(sys.argv[1]) if len(sys.argv) > 1 else 224
T, win, start, length = 65537, 4096, 640, 47360
g = torch.Generator().manual_seed(0)
t = torch.arange(T, dtype=torch.float32)
x = (torch.randn(items, T, generator=g) * torch.exp(-t / 9000.0)).to(dev)
xt = (torch.randn(items, T, generator=g) * torch.exp(-t / 7000.0)).to(dev)
x2 = x.view(-1, 2, T).transpose(1, 2).contiguous()
P = ops.stft_power(xt, win)
Tdb, sabs = ops.edr_target(P)
Tc = ops.edc_target(xt, start, length)
maskw = ((torch.rand(length, generator=g) < 0.5).float() / (0.5 * length)).to(dev)
rows = torch.arange(items, device=dev)


def timed(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / it


def fused(train=True):
    return ops.decay_items_fwd(x2, items, win, Tdb, sabs, rows, None, 1.0, start, length, Tc, maskw, 1.0 / items, 10.0, train)


def unfused():
    Pa = ops.stft_power_pairs(x2, items, win)
    ops.edr_loss(Pa, Tdb, sabs, None, 1.0, True, rows=rows, defer=True)
    ops.edc_loss_pairs(x2, items, start, length, Tc, maskw, 1.0 / items, 10.0, True, rows=rows)


print(f"items {items}: fused {timed(fused):.1f} us   fused (no grad) {timed(lambda: fused(False)):.1f} us   "
      f"unfused chain (one stream) {timed(unfused):.1f} us")
gP, part, edc, dxe = fused()


def bwd():
    gg = ops.stft_power_pairs_bwd_planar(x2, items, win, gP, 0)
    ops.stft_power_pairs_bwd_planar(x2, items, win, gP, 1, out=gg, base=dxe, start=start)


print(f"STFT adjoint (two launches, planar base): {timed(bwd):.1f} us")


if '--stamps' in sys.argv:
    import ctypes
    from diffgfdn_amd import _lib
    lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), 'libdecay_stamps.so'))
    name, (res, args) = 'gfdn_decay_items_fwd', _lib.SIGNATURES['gfdn_decay_items_fwd']
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    gP = torch.empty((items, 32, 2049), device=dev); part = torch.empty(items, device=dev)
    edc = torch.empty(items, device=dev); dxe = torch.empty((items, length), device=dev)
    for _ in range(3):
        rc = fn(x2.data_ptr(), T, T, items, win, Tdb.data_ptr(), sabs.data_ptr(), rows.data_ptr(), None, 1.0, start, length,
                Tc.data_ptr(), maskw.data_ptr(), 1.0 / items, 10.0, 1, gP.data_ptr(), part.data_ptr(), edc.data_ptr(),
                dxe.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
    st = (ctypes.c_longlong * 64)()
    assert lib.gfdn_decay_stamps(st) == 0
    st = list(st)
    t0 = st[0]
    names = {0: 'start', 40: 'rounds done', 41: 'losses out', 42: 'EDR prefix done', 43: 'EDC adjoint done'}
    for r in range(4):
        names.update({1 + 8 * r: f'r{r} top', 2 + 8 * r: f'r{r} windowed (x landed)', 3 + 8 * r: f'r{r} fft done',
                      4 + 8 * r: f'r{r} split + P rows', 5 + 8 * r: f'r{r} EDR columns', 6 + 8 * r: f'r{r} EDC tile'})
    prev = t0
    for k in sorted(names):
        print(f"{names[k]:28s} {(st[k] - t0) / 100.0:8.2f} us   (+{(st[k] - prev) / 100.0:6.2f})")
        prev = st[k]
