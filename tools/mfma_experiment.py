"""BASELINE.json configs[4]: N = 32 delay lines (4 groups x 8), "fp32 vs bf16 feedback-matmul on MFMA".

The feedback multiply only becomes a matrix-core workload if the per-bin resolvent P_k = (D_k / gamma - A)^-1 is formed
explicitly and contracted with the receiver gains, as the reference does (feedback_loop.py:389-391, model.py:615-619).
This script measures that formulation on the MI355X -- C (32 receivers x 32 lines) . P_k (32 x 32) per bin on
v_mfma_f32_32x32x16_bf16 and on v_mfma_f32_32x32x2_f32 -- against the product path (per-bin float32 SOLVE + output
stage, which never forms P), for K = 65 537 bins, and reports for each the time and the deviation of H from a complex128
evaluation next to the north star's 1e-4 bar.      usage: python tools/mfma_experiment.py  ->  one JSON object"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffgfdn_amd import _lib, hip_ops as ops

dev = 'cuda'
K, G, n, B = 65537, 4, 8, 32
N = G * n
g = torch.Generator().manual_seed(0)
fs = 32000.0
z = torch.polar(torch.ones(K, dtype=torch.float64), np.pi * torch.arange(K, dtype=torch.float64) / (K - 1))
delays = torch.tensor(np.sort(np.random.RandomState(0).choice(np.arange(641, 1601), N, replace=False)), dtype=torch.float64)
T60 = torch.linspace(0.3, 1.5, G, dtype=torch.float64).repeat_interleave(n)
gamma = 10 ** (-3 * delays / (fs * T60))
X = torch.triu((2 * torch.rand(G, n, n, generator=g, dtype=torch.float64) - 1) / np.sqrt(n), 1)
Q = torch.linalg.matrix_exp(X - X.transpose(1, 2))
A = torch.block_diag(*(Q @ Q))                                               # zero coupling: blockdiag(Q_g Q_g)
b = (2 * torch.randn(N, generator=g, dtype=torch.float64) - 1) / N
c = (2 * torch.randn(N, generator=g, dtype=torch.float64) - 1) / N
rg = 2 * torch.rand(B, G, generator=g, dtype=torch.float64) - 1
C = rg.repeat_interleave(n, dim=1) * c[None, :]                              # (B, N) output gains per receiver

# complex128 reference, chunked over the bins (the (K, N, N) tensor is 1 GB in complex128)
H_ref = torch.empty(B, K, dtype=torch.complex128)
P_parts = []
Ad, gd, dd = A.to(dev).to(torch.complex128), gamma.to(dev), delays.to(dev)
for k0 in range(0, K, 8192):
    zk = z[k0:k0 + 8192].to(dev)
    D = torch.diag_embed(zk[:, None] ** dd[None, :] / gd[None, :])
    P = torch.linalg.inv(D - Ad[None])
    P_parts.append(P.to(torch.complex64))
    H_ref[:, k0:k0 + 8192] = ((C.to(dev).to(torch.complex128) @ P) @ b.to(dev).to(torch.complex128)).T.cpu()
P64 = torch.cat(P_parts).contiguous()                                        # (K, N, N) complex64 = 537 MB
scale = float(H_ref.abs().max())


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


lib = _lib.load()
Cf, bf = C.float().to(dev).contiguous(), b.float().to(dev).contiguous()
H = torch.empty(B, K, dtype=torch.complex64, device=dev)
out = {'config': f'N = {N} ({G} groups x {n}), B = {B} receivers, K = {K} bins', 'bar': 1e-4}
for name, flag in (('bf16_mfma_contraction', 1), ('f32_mfma_contraction', 0)):
    run = lambda: _lib.check(lib.gfdn_exp_contract_mfma(P64.data_ptr(), K, Cf.data_ptr(), bf.data_ptr(), flag, H.data_ptr(),
                                                        torch.cuda.current_stream().cuda_stream), name)
    us = timed(run)
    err = float((H.cpu().to(torch.complex128) - H_ref).abs().max()) / scale
    out[name] = {'us': us, 'H_max_rel_err': err, 'passes_1e-4': err < 1e-4,
                 'reads_P_MB': P64.numel() * 8 / 1e6, 'GBs': P64.numel() * 8 / us / 1e3,
                 'note': 'P (K, N, N) must exist first: the explicit inverse is not included in the time'}
# the product path: float32 per-bin solve (8 x 8 blocks, thread per system) + output stage; P never exists
turns, _ = ops.zprep(z.to(dev))
QQ = (Q @ Q).float().to(dev)
ig = (1.0 / gamma).float().to(dev)
dl = delays.float().to(dev)
rgf = rg.float().to(dev)
cf = c.float().to(dev)


def product():
    Y = ops.solve_fwd(turns, None, QQ, dl, ig, bf)
    return ops.compose_fwd(Y, cf, rgf, n)


us = timed(product)
Hp = product()
err = float((Hp.cpu().to(torch.complex128) - H_ref).abs().max()) / scale
out['f32_solve_product_path'] = {'us': us, 'H_max_rel_err': err, 'passes_1e-4': err < 1e-4,
                                 'note': 'gfdn_solve_fwd (thread-per-system 8 x 8 elimination) + gfdn_compose_fwd'}
us_inv = timed(lambda: torch.linalg.inv(torch.diag_embed(z[:4096].to(dev)[:, None] ** dd[None, :] / gd[None, :]).to(torch.complex64)
                                        - Ad[None].to(torch.complex64)), it=5)
out['explicit_inverse_c64_torch_us_per_65537_bins'] = us_inv * K / 4096
print(json.dumps(out, indent=1))
