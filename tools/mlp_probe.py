"""Time variants of the MLP kernels (diagnostic): python tools/mlp_probe.py  (variants built by tools/build_mlp_variants.sh)"""
import ctypes, glob, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = ctypes.c_void_p
dev = torch.device('cuda', 0)
B, F, H, nh, G = 32, 20, 16, 5, 4
torch.manual_seed(0)
pos = torch.rand(838, 3, dtype=torch.float64, device=dev)
rows = torch.randint(0, 838, (B,), device=dev)
freq = (torch.exp(torch.linspace(0, 3.4657, F, device=dev)) * 3.14159265).float()
def p(t): return None if t is None else ctypes.c_void_p(t.data_ptr())
for so in sorted(glob.glob(os.path.join(os.path.dirname(__file__), '_probe', 'mlp_*.so'))):
    lib = ctypes.CDLL(so)
    lib.gfdn_mlp_param_count.restype = ctypes.c_size_t
    lib.gfdn_mlp_param_count.argtypes = [ctypes.c_int] * 4
    lib.gfdn_mlp_bwd_work_bytes.restype = ctypes.c_size_t
    lib.gfdn_mlp_bwd_work_bytes.argtypes = [ctypes.c_int] * 5
    n = lib.gfdn_mlp_param_count(F, H, nh, G)
    w = (0.2 * torch.randn(n, device=dev)).float()
    gains = torch.empty(B, G, device=dev); xhat = torch.empty(B, nh + 1, H, device=dev); rstd = torch.empty(B, nh + 1, device=dev)
    gg = torch.randn(B, G, device=dev); gw = torch.empty_like(w)
    work = torch.empty(lib.gfdn_mlp_bwd_work_bytes(B, F, H, nh, G) // 4, device=dev)
    lib.gfdn_mlp_gains_fwd.argtypes = [P, P, P, P] + [ctypes.c_int] * 5 + [ctypes.c_float] * 2 + [P] * 4
    lib.gfdn_mlp_gains_bwd.argtypes = [P, P, P, P] + [ctypes.c_int] * 5 + [ctypes.c_float] * 2 + [P] * 7
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def fwd(): assert lib.gfdn_mlp_gains_fwd(p(pos), p(rows), p(freq), p(w), B, F, H, nh, G, -1.0, 1.0, p(gains), p(xhat), p(rstd), st) == 0
    def bwd(): assert lib.gfdn_mlp_gains_bwd(p(pos), p(rows), p(freq), p(w), B, F, H, nh, G, -1.0, 1.0, p(gains), p(xhat), p(rstd), p(gg), p(gw), p(work), st) == 0
    res = []
    for fn in (fwd, bwd):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500): fn()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 500 * 1e6)
    print(f"{os.path.basename(so):28s} fwd {res[0]:6.1f} us  bwd(+reduce) {res[1]:6.1f} us  gains[0]={gains[0].tolist()}", flush=True)
