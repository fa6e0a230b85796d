"""Where does the float32 noise of dL/dM come from?  (diagnostic, GPU)

Runs ONE explicit bank step of one band at the bench's size (K = 65 537, 32 receivers), captures the inputs of the
output stage's adjoint (dL/dH as the loss side produced it, the saved transfer functions, gains, filter) and compares
the records path (gfdn_tf_compose_bwd -> gfdn_tf_coefs_bwd: dL/dQQ, dL/db, dL/dc) against a complex128 autograd
evaluation of the SAME linear functional Re<dL/dH, H(QQ, b, c)> through per-bin solves -- the error of this stage alone,
with the loss side's error factored out.      usage: python tools/grad_stage_probe.py"""
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
_, sds, bank, trainer, splits = bench.build_bank_workload(dev, 1234, (500.0,), 64, max_epochs=1,
                                                          train_dir='/tmp/gfdn_probe/t')
cap = {}
orig_bwd, orig_oc, orig_energy = ops.tf_compose_bwd, ops.tf_ortho_coefs, ops.tf_energy


def spy_bwd(turns, logr, coef, delays, n, rgain, gH, Ts, filt, nb, **kw):
    cap.update(turns=turns.clone(), coef=coef.clone(), delays=delays.clone(), n=n, rgain=rgain.clone(), gH=gH.clone(),
               Ts=Ts.clone(), filt=None if filt is None else filt.clone(), nb=nb)
    return orig_bwd(turns, logr, coef, delays, n, rgain, gH, Ts, filt, nb, **kw)


def spy_oc(M, ig, b, c, *a, **kw):
    out = orig_oc(M, ig, b, c, *a, **kw)
    cap.update(ig=ig.clone(), b_old=b.clone(), c_old=c.clone(), QQ=out[1].clone(), M=M.clone(), Q=out[0].clone())
    return out


def spy_energy(*a, **kw):
    out = orig_energy(*a, **kw)
    cap['scale'] = out[1].clone()
    return out


ops.tf_compose_bwd, ops.tf_ortho_coefs, ops.tf_energy = spy_bwd, spy_oc, spy_energy
rows = sds.global_rows([splits[0][0][:bench.BATCH]])
batch = sds.collate(rows)
start, length = trainer._decay_window(bench.K)
maskw = torch.full((length,), 1.0 / (bench.BATCH * length), device=dev)
trainer._fused.run(batch, maskw, 1.0, normalize_first=True, train=True, opt_step=False)
torch.cuda.synchronize()
ops.tf_compose_bwd, ops.tf_ortho_coefs, ops.tf_energy = orig_bwd, orig_oc, orig_energy
gM_step = trainer._fused.g_M.detach().clone()            # dL/dM of the whole step (all loss terms), from the flat buffer

n, nb = cap['n'], cap['nb']
G = cap['QQ'].shape[0] // nb
s = cap['scale'].double()
rs = s.sqrt().repeat_interleave(n)
z = torch.polar(torch.ones_like(cap['turns']), 2 * np.pi * cap['turns'])              # slot-ordered grid, complex128
QQ = cap['QQ'].double().requires_grad_()
bp = (cap['b_old'].double() * rs).requires_grad_()
cp = (cap['c_old'].double() * rs).requires_grad_()
ig, delays = cap['ig'].double(), cap['delays'].double()
T = []
for q in range(G):
    sl = slice(q * n, (q + 1) * n)
    D = torch.diag_embed(z[:, None] ** delays[sl][None, :] * ig[sl][None, :])
    y = torch.linalg.solve(D - QQ[q].to(torch.complex128)[None], bp[sl].to(torch.complex128)[None, :, None].expand(z.numel(), n, 1))
    T.append((cp[sl].to(torch.complex128)[None, :] * y[..., 0]).sum(-1))
T = torch.stack(T, dim=0)                                                              # (G, Ku)
H = torch.einsum('bg,gk->bk', cap['rgain'].double().to(torch.complex128), T)
if cap['filt'] is not None:
    H = H * cap['filt'].reshape(1, -1).to(torch.complex128)
W = cap['gH'].to(torch.complex128)
L = (H.real * W.real + H.imag * W.imag).sum()
L.backward()
print('saved transfer functions vs complex128:', float((cap['Ts'].to(torch.complex128) - T.detach()).abs().max() / T.detach().abs().max()))
grec = ops.tf_compose_bwd(cap['turns'], None, cap['coef'], cap['delays'], n, cap['rgain'], cap['gH'], cap['Ts'], cap['filt'], nb)
gA, _, gb, gc = ops.tf_coefs_bwd(cap['QQ'], cap['ig'], grec, bp.detach().float(), cp.detach().float())


def dev_rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


print('records path vs complex128 autograd on the same dL/dH (max-norm, relative to the largest entry):')
print('  dL/dQQ', f'{dev_rel(gA, QQ.grad):.2e}', '  dL/db', f'{dev_rel(gb, bp.grad):.2e}', '  dL/dc', f'{dev_rel(gc, cp.grad):.2e}')
grg = ops.tf_gain_grad(cap['Ts'], cap['gH'], G, cap['filt'], nb)
ref_rg = torch.einsum('bk,gk->bg', W.real, T.detach().real * 0 + (T.detach() * (cap['filt'].reshape(1, -1).to(torch.complex128) if cap['filt'] is not None else 1)).real) \
    + torch.einsum('bk,gk->bg', W.imag, (T.detach() * (cap['filt'].reshape(1, -1).to(torch.complex128) if cap['filt'] is not None else 1)).imag)
print('  dL/drgain', f'{dev_rel(grg, ref_rg):.2e}')
# conditioning of the sums: |sum over bins| against sum over bins of |terms| for dL/dQQ
print('  |dL/dQQ| largest entry', float(QQ.grad.abs().max()), ' |dL/db| ', float(bp.grad.abs().max()), ' |dL/dc| ', float(cp.grad.abs().max()))
print('  |dL/dM| largest entry (whole step, after the expm adjoint + skew projection)', float(gM_step.abs().max()),
      ' -> a 1e-6 error of dL/dQQ relative to ITS largest entry is',
      f'{1e-6 * float(QQ.grad.abs().max()) / float(gM_step.abs().max()):.1e}', 'of the largest entry of dL/dM')


# ---- round 4: which part of the records path carries the error?  (i) the kernel's records against float64 sums of the same
# per-bin terms; (ii) the EXACT records rounded to float32 through the kernel's float64 map -> dL/dQQ: the floor that
# float32 STORAGE of the records sets; (iii) the same through the expm adjoint: the stage's share of the error of dL/dM
with torch.no_grad():
    turns = cap['turns'].double()
    rec_exact = torch.zeros((G, 32), dtype=torch.float64, device=dev)
    coef64 = cap['coef'].double()
    filt = cap['filt'].reshape(-1).to(torch.complex128) if cap['filt'] is not None else None
    for q in range(G):
        m = delays[q * n:(q + 1) * n]
        e1 = [torch.polar(torch.ones_like(turns), 2 * np.pi * ((turns * m[i]) % 1.0)) for i in range(n)]
        eS = []
        for S in range(16):
            v = torch.ones_like(e1[0])
            for i in range(n):
                if (S >> i) & 1:
                    v = v * e1[i]
            eS.append(v)
        eS = torch.stack(eS)                                               # (16, Ku)
        den = (coef64[q, 16:32, None] * eS).sum(0)
        acc = (cap['rgain'][:, q].double().to(torch.complex128)[:, None] * W).sum(0)
        if filt is not None:
            acc = acc * filt.conj()
        u = acc * (1.0 / den).conj()
        v = u * T.detach()[q].conj()
        rec_exact[q, :16] = (u[None, :] * eS.conj()).real.sum(1)
        rec_exact[q, 16:] = -(v[None, :] * eS.conj()).real.sum(1)
        rec_exact[q, 15] = 0.0
    gk = grec.double().reshape(G, 32)
    scale_rec = rec_exact.abs().max(dim=1, keepdim=True).values
    print('(i) kernel records vs float64 sums of the same terms: worst entry relative to the block\'s largest record',
          f'{float(((gk - rec_exact).abs() / scale_rec).max()):.2e}',
          '; float32 rounding of the exact records alone:',
          f'{float(((rec_exact.float().double() - rec_exact).abs() / scale_rec).max()):.2e}')
    gA_x, _, gb_x, gc_x = ops.tf_coefs_bwd(cap['QQ'], cap['ig'], rec_exact.float().contiguous(), bp.detach().float(),
                                           cp.detach().float())
    print('(ii) exact records rounded to float32 -> float64 map: dL/dQQ', f'{dev_rel(gA_x, QQ.grad):.2e}',
          ' dL/db', f'{dev_rel(gb_x, bp.grad):.2e}', ' dL/dc', f'{dev_rel(gc_x, cp.grad):.2e}')

# (iii) through the expm adjoint (dL/dM of THIS linear functional only)
M64 = cap['M'].double().requires_grad_()
Sk = torch.triu(M64, 1)
Qm = torch.linalg.matrix_exp(Sk - Sk.transpose(-1, -2))
(Qm @ Qm * QQ.grad).sum().backward()
gM_ref = M64.grad
gM_k = ops.ortho_bwd(cap['M'], None, gA.contiguous(), cap['Q'])
gM_x = ops.ortho_bwd(cap['M'], None, gA_x.contiguous(), cap['Q'])
gM_f32in = ops.ortho_bwd(cap['M'], None, QQ.grad.float().contiguous(), cap['Q'])
print('(iii) dL/dM of this functional, relative to its largest entry', float(gM_ref.abs().max()), ': kernel records',
      f'{dev_rel(gM_k, gM_ref):.2e}', '; exact records rounded to float32', f'{dev_rel(gM_x, gM_ref):.2e}',
      '; exact dL/dQQ rounded to float32', f'{dev_rel(gM_f32in, gM_ref):.2e}')
