"""Stage-by-stage wall/GPU time of one training step (diagnostic; not part of the product)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from diffgfdn_amd import hip_ops as ops
from diffgfdn_amd.functional import OutputStage
from diffgfdn_amd.losses import decay_losses
from diffgfdn_amd.colorless_losses import group_spectral_loss

dev = torch.device('cuda', 0)
room, data, net, trainer, train_idx, filt, delays = bench.build_workload(dev, 1234, int(os.environ.get('R', 838)))
sel = train_idx[:32]

def timed(name, fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name:34s} {dt*1e3:8.3f} ms", flush=True)
    return r

batch = timed('collate(lean)', lambda: data.collate(sel, lean=True))
timed('normalize', lambda: trainer.normalize(batch))
z = batch['z_values']
rg = timed('mlp group_gains', lambda: net.output_scalars.group_gains(batch))
timed('feedback_blocks (expm)', lambda: net.feedback_loop.feedback_blocks())
Y = timed('solve fwd (delay_line_responses)', lambda: net.delay_line_responses(z))
H = timed('output stage', lambda: OutputStage.apply(Y, net.output_gains.reshape(-1), rg.float(), 4, batch['target_early_response'], filt))
timed('draw_mask', lambda: trainer.criterion[1].draw_mask(47360, dev))
Hd = H.detach()
timed('irfft_odd_fwd', lambda: ops.irfft_odd_fwd(Hd, bench.K))
x = ops.irfft_odd_fwd(Hd, bench.K)
timed('stft_power', lambda: ops.stft_power(x, 4096))
edr_t, edc_t = batch['edr_target'], batch['edc_target']
timed('edc_loss', lambda: ops.edc_loss(x, 640, 47360, edc_t[1], None, 1.0, 1.0, True))
P = ops.stft_power(x, 4096)
timed('edr_loss', lambda: ops.edr_loss(P.clone(), edr_t[1], edr_t[2], None, 1.0, True))
gx = torch.zeros_like(x)
timed('stft_power_bwd', lambda: ops.stft_power_bwd(x, 4096, P, gx))
timed('irfft_odd_bwd', lambda: ops.irfft_odd_bwd(gx, bench.K, bench.K))
Hg = H.detach().requires_grad_(True)
timed('decay_losses fwd+grad', lambda: decay_losses(Hg, None, edc_start=640, edc_len=47360, edr_target=(edr_t[1], edr_t[2]), edc_target=edc_t[1], edr_weight=1.0, edc_weight=10.0))
timed('sub_fdn_group_sums', lambda: net.sub_fdn_group_sums(z))
def full_losses():
    trainer.optimizer.zero_grad(set_to_none=True)
    l = trainer._step_losses(batch); return l
timed('_step_losses (fwd)', full_losses)
def fb():
    l = full_losses(); l['_total'].backward()
timed('_step_losses + backward', fb)
timed('optimizer.step', lambda: trainer.optimizer.step())
timed('train_step', lambda: trainer.train_step(batch))
def whole():
    b = data.collate(sel, lean=True); trainer.normalize(b); trainer.train_step(b)
timed('whole step', whole)
