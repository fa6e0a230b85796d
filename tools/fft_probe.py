"""Time variants of the FFT kernels (diagnostic): python tools/fft_probe.py  (variants built by tools/build_fft_variants.sh)"""
import ctypes, glob, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
dev = torch.device('cuda', 0)
B, K, WIN = int(os.environ.get('B', 224)), 65537, 4096
torch.manual_seed(0)
x = torch.randn(B, K, device=dev)
X = torch.randn(B, (K + 1) // 2, dtype=torch.complex64, device=dev)
gP = torch.rand(B, 32, 2049, device=dev)
def p(t): return None if t is None else ctypes.c_void_p(t.data_ptr())
ref = {}
for so in sorted(glob.glob(os.path.join(os.path.dirname(__file__), '_probe', 'fft_*.so'))):
    lib = ctypes.CDLL(so)
    lib.gfdn_bluestein_table_bytes.restype = ctypes.c_size_t; lib.gfdn_bluestein_table_bytes.argtypes = [I]
    lib.gfdn_bluestein_work_bytes.restype = ctypes.c_size_t; lib.gfdn_bluestein_work_bytes.argtypes = [I, I]
    lib.gfdn_bluestein_table_init.argtypes = [I, P]
    lib.gfdn_irfft_odd_stages.argtypes = [P, I, P, P, I, I, P, I, P, I, I, I, P]
    lib.gfdn_stft_power.argtypes = [P, I, I, I, I, P, P, P]
    lib.gfdn_stft_power_bwd.argtypes = [P, I, I, I, I, P, P, P]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    table = torch.empty(lib.gfdn_bluestein_table_bytes(K), dtype=torch.uint8, device=dev)
    assert lib.gfdn_bluestein_table_init(K, p(table)) == 0
    work = torch.empty(lib.gfdn_bluestein_work_bytes(K, B), dtype=torch.uint8, device=dev)
    xo = torch.empty(B, K, device=dev); gX = torch.empty(B, (K + 1) // 2, dtype=torch.complex64, device=dev)
    Pw = torch.empty(B, 32, 2049, device=dev); gx = torch.zeros(B, K, device=dev)
    def stage(adj, stg):
        if adj: return lambda: lib.gfdn_irfft_odd_stages(p(table), K, p(x), None, K, B, p(gX), (K + 1) // 2, p(work), 1, stg, SLOTS, st)
        return lambda: lib.gfdn_irfft_odd_stages(p(table), K, p(X), None, (K + 1) // 2, B, p(xo), K, p(work), 0, stg, SLOTS, st)
    SLOTS = int(os.environ.get('SLOTS', '0'))
    fns = [('col_fwd', stage(0, 1)), ('row', stage(0, 2)), ('col_inv', stage(0, 4)), ('irfft', stage(0, 7)),
           ('a.col_fwd', stage(1, 1)), ('a.row', stage(1, 2)), ('a.col_inv', stage(1, 4)), ('a.irfft', stage(1, 7)),
           ('stft', lambda: lib.gfdn_stft_power(p(x), K, K, B, WIN, p(Pw), None, st)),
           ('stft_bwd', lambda: lib.gfdn_stft_power_bwd(p(x), K, K, B, WIN, p(gP), p(gx), st))]
    res = []
    for name, fn in fns:
        for _ in range(3): assert fn() == 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); res.append(f"{name} {(time.perf_counter() - t0) / 50 * 1e6:6.1f}")
    # correctness against the first variant
    stage(0, 7)(); stage(1, 7)(); gx.zero_(); fns[-2][1](); fns[-1][1](); torch.cuda.synchronize()
    outs = {'x': xo.clone(), 'gX': torch.view_as_real(gX).clone(), 'P': Pw.clone(), 'gx': gx.clone()}
    if not ref: ref = outs
    err = ' '.join(f"{k}:{float((outs[k] - ref[k]).abs().max() / ref[k].abs().max()):.1e}" for k in outs)
    print(f"{os.path.basename(so):16s} " + ' | '.join(res) + '  err ' + err, flush=True)
