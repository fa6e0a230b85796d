"""Where does the float32 deviation of dL/dM enter the LINEAR step?  (diagnostic, GPU; round 5)

One explicit bank step of one band at the bench's size (K = 65 537, 32 receivers) at the parameters, batch and EDC mask of
the CPU oracle's step (bench.cpu_reference_step); the launches' intermediate results are captured and the chain is
re-evaluated in float64 (torch, complex128 per-bin solves, torch.fft) FROM successive hand-over points on:
    V0  every stage on the float32 kernels                                            (what the step does)
    VA  float64 from dL/dtau on     (kernel's gamma -> adjoint transform, records, cofactor map, expm adjoint in float64)
    VB  float64 from the loss gradients on   (kernel's dL/dx rows and summed gradient spectra -> sums over the receivers,
        adjoint STFT, adjoint transform, ... in float64)
each as the deviation of the step's total dL/dM from the oracle's, in max-norm relative to its largest entry -- so the
difference V0 -> VA is what float64 adjoint transforms + records would buy, VA -> VB what float64 receiver sums and a
float64 adjoint STFT would add, and VB is the share of the float32 loss stages (forward transform, STFT, dB stages, scans).
usage: python tests/grad_stage_probe.py   (a diagnostic beside tests/grad_noise_floor.py: it runs the CPU oracle through bench.cpu_reference_step)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import bench                                   # noqa: E402
from diffgfdn_amd import hip_ops as ops        # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
K, WIN, G, NPER = bench.K, bench.WIN, bench.G, bench.NPER
cpu_leg, _sds, _bank, _tr, _splits = bench.build_bank_workload(dev, 1234, (500.0,), 64, max_epochs=1, train_dir='/tmp/gfdn_probe/t')
room, delays, filt = cpu_leg
del _sds, _bank, _tr
filt_np = filt.cpu().numpy().astype(np.complex128)
ref = bench.cpu_reference_step(room, delays, filt_np, steps=0)
gM_ref = torch.tensor(ref['grads0']['M']).reshape(G, NPER, NPER)

cap = {}
names = ('irfft_odd_fwd', 'irfft_odd_pairs_bwd', 'lin_merge_slots', 'edc_lin_one', 'edr_lin_loss_gsum', 'tf_param_grads', 'mlp_gains_fwd',
         'tf_compose_bwd')
orig = {k: getattr(ops, k) for k in names}


def spy(name):
    def f(*a, **kw):
        out = orig[name](*a, **kw)
        cap[name] = (a, kw, out)
        return out
    return f


for k in names:
    setattr(ops, k, spy(k))
losses, grads, tr, bank, sds = bench.hip_step_at(ref['init'], ref['rirs'], ref['pos'], delays, room, filt_np, ref['keep0'], dev,
                                                 want_objects=True)
torch.cuda.synchronize()
for k in names:
    setattr(ops, k, orig[k])


def dev_rel(a, r):
    a, r = a.detach().double().cpu(), r.detach().double().cpu()
    return float((a - r).abs().max() / r.abs().max())


gM0 = torch.tensor(grads['M']).reshape(G, NPER, NPER)
print(f"oracle |dL/dM| largest entry {float(gM_ref.abs().max()):.4f}")
print(f"V0  all stages float32 kernels:                         dL/dM deviation {dev_rel(gM0, gM_ref):.2e}")

# ---- the float64 model of tau_g(M, b, c) = irfft(T_g filt, n = K) at the step's (normalized) parameters
a_pg, kw_pg, _ = cap['tf_param_grads']
QQ_k, ig, grec, b_k, c_k, M_k = a_pg[:6]
grec_sub, gQ, Q_k = kw_pg['grec1'], kw_pg['gQ'], kw_pg['Q']
n = NPER
Ku = (K + 1) // 2
zk = torch.polar(torch.ones(Ku, dtype=torch.float64, device=dev), 2 * np.pi * torch.arange(Ku, dtype=torch.float64, device=dev) / bench.NFFT)
filt64 = tr.subband_filter_freq_resp[0, :Ku].to(torch.complex128)
dl = bank.delays.double().reshape(G, n)
ig64 = bank.inv_gamma.double().reshape(G, n)


def tau_of(M64, b64, c64):
    Sk = torch.triu(M64, 1)
    Qm = torch.linalg.matrix_exp(Sk - Sk.transpose(-1, -2))
    QQ = Qm @ Qm
    taus = []
    for g in range(G):
        D = torch.diag_embed(zk[:, None] ** dl[g][None, :] * ig64[g][None, :])
        y = torch.linalg.solve(D - QQ[g].to(torch.complex128)[None],
                               b64[g].to(torch.complex128)[None, :, None].expand(Ku, n, 1))
        T = (c64[g].to(torch.complex128)[None, :] * y[..., 0]).sum(-1)
        taus.append(torch.fft.irfft(T * filt64, n=K))
    return torch.stack(taus)                                                   # (G, K) float64


M64 = M_k.double().reshape(G, n, n).clone().requires_grad_()
b64 = b_k.double().reshape(G, n)
c64 = c_k.double().reshape(G, n)
tau64 = tau_of(M64, b64, c64)

# main-branch share of the kernel's dL/dM alone (records of the decay losses -> dL/dQQ -> expm adjoint), to swap it out
gM_main_k, _, _ = ops.tf_param_grads(QQ_k, ig, grec, b_k, c_k, M_k, Q=Q_k)
gM_main_k = gM_main_k.reshape(G, n, n).double().cpu()


def variant(L):
    M64.grad = None
    L.backward(retain_graph=True)
    return gM0.double() - gM_main_k + M64.grad.detach().cpu()


# ---- VA: from the kernel's gamma (time order: the three parts the merge launch adds)
a_m, _, gam_slots = cap['lin_merge_slots']
gam_t = a_m[0].double()
for extra in a_m[1:3]:
    if extra is not None:
        gam_t = gam_t + extra.double()
gamma_time = torch.stack([gam_t[g // 2, :, g % 2] for g in range(G)])        # (G, K)
gA = variant((gamma_time * tau64).sum())
print(f"VA  float64 from dL/dtau on (adjoint transform, records):   dL/dM deviation {dev_rel(gA, gM_ref):.2e}")

# ---- VB: from the kernel's loss gradients: dL/dx rows (EDC, window) and the gradient spectra summed over the receivers
(a_e, kw_e, (li_edc, gx)) = cap['edc_lin_one']
rgain = cap['mlp_gains_fwd'][2][0].double()                                    # (items, G)
start, length = tr._decay_window(K)
g_edc = (rgain.t()[:, :, None] * gx.double()[None, :, :]).sum(1)             # (G, L): sum_b gain[b][g] dL/dx_b on the window
Gs = cap['edr_lin_loss_gsum'][2][1]                                            # (nsplit, G, frames, 2049) tiled complex64
Gs = ops.spec_tile(Gs.sum(0) if Gs.dim() == 4 else Gs, inverse=True).to(torch.complex128)
pad = (-K) % (WIN // 2)
taup = torch.nn.functional.pad(tau64, (0, pad))
S64 = torch.stft(taup, WIN, hop_length=WIN // 2, window=torch.hann_window(WIN, dtype=torch.float64, device=dev),
                 center=False, return_complex=True).transpose(-1, -2)           # (G, frames, 2049)
L_B = (g_edc * tau64[:, start:start + length]).sum() + (Gs.real * S64.real + Gs.imag * S64.imag).sum()
gB = variant(L_B)
print(f"VB  float64 from the loss gradients on (sums, adjoint STFT, ...): dL/dM deviation {dev_rel(gB, gM_ref):.2e}")

# ---- stage outputs against float64 evaluations of the same stage on the kernel's own inputs
with torch.no_grad():
    gam64 = g_edc.new_zeros((G, K))
    gam64[:, start:start + length] = g_edc
    M64.grad = None
taup2 = taup.detach().clone().requires_grad_()
S2 = torch.stft(taup2, WIN, hop_length=WIN // 2, window=torch.hann_window(WIN, dtype=torch.float64, device=dev), center=False,
                return_complex=True).transpose(-1, -2)
(Gs.real * S2.real + Gs.imag * S2.imag).sum().backward()
gam64 = gam64 + taup2.grad[:, :K]
print(f"    stage: gamma (receiver sums + adjoint STFT) kernel vs float64 on the same dL/dx, Gsum: {dev_rel(gamma_time, gam64):.2e}")
a_t, kw_t, gHg = cap['irfft_odd_pairs_bwd']
X = torch.zeros((G, Ku), dtype=torch.complex128, device=dev, requires_grad=True)
(gamma_time * torch.fft.irfft(X, n=K)).sum().backward()
order = ops.irfft_slot_order(K, dev)
bins, conj = order
gnat = X.grad                                   # torch's convention: dL/dRe + i dL/dIm
gslot = torch.cat([gnat[:, :1], torch.where(conj[None, :], gnat[:, bins].conj(), gnat[:, bins])], dim=1)
print(f"    stage: adjoint transform kernel vs float64 on the same gamma (slot order, largest entry): {dev_rel(gHg.to(torch.complex128).abs(), gslot.abs()):.2e} (moduli)")
print("losses vs oracle:", {k: f"{abs(losses[k] - ref['first'][k]) / abs(ref['first'][k]):.1e}" for k in losses})

# ---- the two branches separately: the decay-loss branch (through dL/dtau) and the colorless branch (spectral loss on the raw
# sub-FDN responses of the K uniform bins + sparsity of the LAST group's Q, trainer.py:298-308), each against float64
gM_col_k = gM0.double() - gM_main_k
M64.grad = None
L_B.backward(retain_graph=True)
gM_main_64 = M64.grad.detach().cpu().clone()
zK = torch.polar(torch.ones(K, dtype=torch.float64, device=dev), 2 * np.pi * torch.arange(K, dtype=torch.float64, device=dev) / bench.NFFT)
Mc = M_k.double().reshape(G, n, n).clone().requires_grad_()
loss_c = 0.0
for g in range(G):
    D = torch.diag_embed(zK[:, None] ** dl[g][None, :])
    y = torch.linalg.solve(D - Mc[g].to(torch.complex128)[None], b64[g].to(torch.complex128)[None, :, None].expand(K, n, 1))
    S = (c64[g].to(torch.complex128)[None, :] * y[..., 0]).sum(-1)
    d = S.abs() - 1.0
    loss_c = loss_c + torch.where(d > 1.0, d ** 4, d ** 2).mean()              # (asymmetric: exponent 4 where |S| - 1 > 1)
Skc = torch.triu(Mc[G - 1], 1)
Ql = torch.linalg.matrix_exp(Skc - Skc.t())
sparsity = -(Ql.abs().sum() - n * np.sqrt(n)) / (n * (np.sqrt(n) - 1))
(1.0 * loss_c + 2.0 * sparsity).backward()
gM_col_64 = Mc.grad.detach().cpu()
big = float(gM_ref.abs().max())
print(f"    branch: decay losses  kernel vs float64-from-the-loss-gradients: {float((gM_main_k - gM_main_64).abs().max()) / big:.2e} of the total's largest entry"
      f" (branch's own largest entry {float(gM_main_64.abs().max()):.3f})")
print(f"    branch: colorless     kernel vs float64:                          {float((gM_col_k - gM_col_64).abs().max()) / big:.2e} of the total's largest entry"
      f" (branch's own largest entry {float(gM_col_64.abs().max()):.3f})")
print(f"    float64 decay branch (from the kernel's loss gradients) + float64 colorless branch vs oracle: "
      f"{dev_rel(gM_main_64 + gM_col_64, gM_ref):.2e}")
print(f"    kernel decay branch + float64 colorless branch vs oracle: {dev_rel(gM_main_k + gM_col_64, gM_ref):.2e}")


# ---- the WHOLE decay branch in float64 from the parameters (VD), and the same with the loss gradients taken at the kernel's
# float32 time signals (VE: the float32 rounding of the forward transforms -- the band's group signals and the dataset's
# transformed direct paths -- is the only float32 stage)
rows = sds.global_rows([list(range(bench.BATCH))])
rows_t = torch.as_tensor(rows, device=dev)
B = bench.BATCH
Et = sds.early_rir_mag_response[rows_t][:, :Ku].to(torch.complex128)
Ht = sds.datasets[0].rir_mag_response[rows_t][:, :Ku].to(torch.complex128)
rg64 = rgain                                                                   # (B, G) float64 (kernel's network output)
maskw = torch.zeros(length, dtype=torch.float64, device=dev)
maskw[ref['keep0'].reshape(-1).to(dev)] = 1.0 / (B * ref['keep0'].numel())
win64 = torch.hann_window(WIN, dtype=torch.float64, device=dev)
EPS = 1.1920928955078125e-07


def db64(p):
    return torch.clamp(10.0 * torch.log10(p.abs() + EPS), min=-200.0)


def edr_db(x):
    S = torch.stft(torch.nn.functional.pad(x, (0, (-K) % (WIN // 2))), WIN, hop_length=WIN // 2, window=win64, center=False,
                   return_complex=True)                                        # (B, 2049, frames)
    P = S.real ** 2 + S.imag ** 2
    return db64(torch.flip(torch.cumsum(torch.flip(P, [-1]), -1), [-1]))


def edc_db(x):
    w = x[:, start:start + length]
    return db64(torch.flip(torch.cumsum(torch.flip(w * w, [-1]), -1), [-1]))


t_rir = torch.fft.irfft(Ht, n=K)
T_edr, T_edc = edr_db(t_rir), edc_db(t_rir)


def decay_losses(x):
    a_edr, a_edc = edr_db(x), edc_db(x)
    edr = ((T_edr - a_edr).abs().sum(dim=(-1, -2)) / T_edr.abs().sum(dim=(-1, -2))).sum()
    edc = (maskw[None, :] * (T_edc - a_edc).abs()).sum()
    return 1.0 * edr + 10.0 * edc, edr, edc


xd64 = torch.fft.irfft(Et * filt64[None, :], n=K)
x64 = xd64 + rg64 @ tau64                                                      # (B, K)
L_D, edr64, edc64 = decay_losses(x64)
gD = variant(L_D)
print(f"VD  the whole decay branch in float64 from the parameters:      dL/dM deviation {dev_rel(gD, gM_ref):.2e}"
      f"   (float64 losses vs oracle: EDR {abs(float(edr64) - ref['first']['edr_loss']) / ref['first']['edr_loss']:.1e},"
      f" EDC {abs(10 * float(edc64) - ref['first']['edc_loss']) / ref['first']['edc_loss']:.1e})")
tau_k2 = cap['irfft_odd_fwd'][2]                                               # (G / 2, K, 2) float32, the kernel's group signals
tau_k = torch.stack([tau_k2[g // 2, :, g % 2] for g in range(G)]).double()
xd_k = sds.direct_time(tr.subband_filter_freq_resp, K)[rows_t].double()
x32 = (xd_k + rg64 @ tau_k).detach().requires_grad_()
L32, _, _ = decay_losses(x32)
g_at_x32, = torch.autograd.grad(L32, x32)
gE = variant((g_at_x32 * x64).sum())
print(f"VE  float64 everywhere, loss gradients taken at the kernel's float32 signals: dL/dM deviation {dev_rel(gE, gM_ref):.2e}")
print(f"    the kernel's signals against float64: group signals {dev_rel(tau_k, tau64.detach()):.2e} of their largest sample,"
      f" direct paths {dev_rel(xd_k, xd64):.2e}; on the last 10 % of the EDC window relative to the largest sample THERE:"
      f" {float((tau_k - tau64.detach())[:, start + int(0.9 * length):start + length].abs().max() / tau64.detach()[:, start + int(0.9 * length):start + length].abs().max()):.2e}")
xmix = (xd64 + rg64 @ tau_k).detach().requires_grad_()
Lm, _, _ = decay_losses(xmix)
gm, = torch.autograd.grad(Lm, xmix)
gF = variant((gm * x64).sum())
print(f"VF  as VE with float64 direct paths (float32 group signals only): dL/dM deviation {dev_rel(gF, gM_ref):.2e}")
xmix2 = (xd_k + rg64 @ tau64.detach()).detach().requires_grad_()
Lm2, _, _ = decay_losses(xmix2)
gm2, = torch.autograd.grad(Lm2, xmix2)
gG = variant((gm2 * x64).sum())
print(f"VG  as VE with float64 group signals (float32 direct-path store only): dL/dM deviation {dev_rel(gG, gM_ref):.2e}")
print("---- the same against the ALL-FLOAT64 evaluation VD (complex128 resolvent, float64 transforms and loss stages) instead of the"
      " oracle, which follows the reference's casts (complex64 resolvent, float32 EDR buffer: feedback_loop.py:389-391, losses.py:566-567):")
print(f"    oracle (reference arithmetic) vs all-float64: {dev_rel(gM_ref, gD):.2e}")
print(f"    V0 float32 kernels            vs all-float64: {dev_rel(gM0, gD):.2e}")
print(f"    VB float64 from the loss gradients on         : {dev_rel(gB, gD):.2e}")
print(f"    VE loss gradients at the kernel's signals     : {dev_rel(gE, gD):.2e}")
print(f"    VF float32 group signals only                 : {dev_rel(gF, gD):.2e}")
print(f"    VG float32 direct-path store only             : {dev_rel(gG, gD):.2e}")
# the reference's complex64 cast of the transfer functions alone: T rounded to complex64 before the (float64) transform
T32 = torch.fft.rfft(tau64.detach(), n=K)[:, :Ku].to(torch.complex64).to(torch.complex128)
tau_c64 = torch.fft.irfft(T32, n=K)
xc = (xd64 + rg64 @ tau_c64).detach().requires_grad_()
Lc, _, _ = decay_losses(xc)
gc_, = torch.autograd.grad(Lc, xc)
print(f"    float64 with the group spectra rounded to complex64 (the reference's cast of P, model.py:583-619) : {dev_rel(variant((gc_ * x64).sum()), gD):.2e}")
