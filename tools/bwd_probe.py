import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nt = int(os.environ.get('NT', '0'))
if nt: torch.set_num_threads(nt)
import bench
from diffgfdn_amd.functional import OutputStage
from diffgfdn_amd.losses import decay_losses
from diffgfdn_amd.colorless_losses import group_spectral_loss, sparsity_loss
dev = torch.device('cuda', 0)
print('torch threads', torch.get_num_threads(), flush=True)
room, data, net, trainer, train_idx, filt, delays = bench.build_workload(dev, 1234, 128)
sel = train_idx[:32]
batch = data.collate(sel, lean=True)
z = batch['z_values']
edr_t, edc_t = batch['edr_target'], batch['edc_target']
def timed(name, fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name:40s} {(time.perf_counter()-t0)/n*1e3:8.3f} ms", flush=True)
def zg(): trainer.optimizer.zero_grad(set_to_none=True)
def f_sparsity():
    zg(); fl = net.feedback_loop; l = sparsity_loss()(fl.ortho_param(fl.M[3])); l.backward()
def f_spectral():
    zg(); S, _ = net.sub_fdn_group_sums(z); l = group_spectral_loss(S, True); l.backward()
def f_decay():
    zg(); rg = net.output_scalars.group_gains(batch); Y = net.delay_line_responses(z)
    H = OutputStage.apply(Y, net.output_gains.reshape(-1), rg.float(), 4, batch['target_early_response'], filt)
    t, _, _ = decay_losses(H, None, edc_start=640, edc_len=47360, edr_target=(edr_t[1], edr_t[2]), edc_target=edc_t[1], edc_weight=10.0)
    t.backward()
def f_decay_nomlp():
    zg(); rg = torch.ones(32, 4, device=dev); Y = net.delay_line_responses(z)
    H = OutputStage.apply(Y, net.output_gains.reshape(-1), rg, 4, batch['target_early_response'], filt)
    t, _, _ = decay_losses(H, None, edc_start=640, edc_len=47360, edr_target=(edr_t[1], edr_t[2]), edc_target=edc_t[1], edc_weight=10.0)
    t.backward()
def f_solve_only():
    zg(); Y = net.delay_line_responses(z); (Y.real.sum()).backward()
def f_expm_only():
    zg(); A = net.feedback_loop.feedback_blocks(); A.sum().backward()
def f_mlp_only():
    zg(); rg = net.output_scalars.group_gains(batch); rg.sum().backward()
timed('sparsity fwd+bwd', f_sparsity)
timed('expm blocks fwd+bwd', f_expm_only)
timed('mlp fwd+bwd', f_mlp_only)
timed('solve fwd+bwd', f_solve_only)
timed('spectral fwd+bwd', f_spectral)
timed('decay (no mlp) fwd+bwd', f_decay_nomlp)
timed('decay fwd+bwd', f_decay)
timed('draw_mask', lambda: trainer.criterion[1].draw_mask(47360, dev))
def whole():
    b = data.collate(sel, lean=True); trainer.normalize(b); trainer.train_step(b)
timed('whole step', whole, 20)
