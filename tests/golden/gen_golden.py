"""Generate golden vectors from the REFERENCE implementation (build-container only).

    python tests/golden/gen_golden.py

Imports /root/reference/src/diff_gfdn through tests/golden/ref_import.py, builds seeded
models / batches, runs the reference forward, losses, backward and one Adam step, and
writes small .npz fixtures (inputs + expected outputs) next to this file.  The fixtures
are DATA; no reference source text is stored.  tests/test_oracle_golden.py pins the
oracle to them, tests/test_gpu_parity.py pins the HIP path to them.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.install()

from diff_gfdn.colorless_fdn.losses import amse_loss, mse_loss, sparsity_loss  # noqa: E402
from diff_gfdn.config.config import (CouplingMatrixType, FeedbackLoopConfig,  # noqa: E402
                                     OutputFilterConfig, TrainerConfig)
from diff_gfdn.feedback_loop import FeedbackLoop  # noqa: E402
from diff_gfdn.losses import directional_edc_loss, edc_loss, edr_loss  # noqa: E402
from diff_gfdn.model import (DiffDirectionalFDNVarReceiverPos, DiffGFDNSinglePos,  # noqa: E402
                             DiffGFDNVarReceiverPos, DiffGFDNVarSourceReceiverPos)
from diff_gfdn.trainer import VarReceiverPosTrainer  # noqa: E402
import diff_gfdn.losses as ref_losses  # noqa: E402
import spatial_sampling.model as ss_model  # noqa: E402


def c2np(t):
    return t.detach().cpu().numpy().copy()


def prime_delays(n, lo=20, hi=50, seed=0):
    """Small distinct primes; the reference rule (config.py:262-279) scaled down."""
    import sympy as sp
    primes = np.array(list(sp.primerange(lo, hi)), dtype=np.int32)
    rng = np.random.RandomState(seed)
    sel = primes[rng.permutation(len(primes))][:n - 1]
    return [int(v) for v in sel] + [int(sp.nextprime(hi))]


def synth_batch(B, nfft, fs, G, T, seed, T60=None):
    """Synthetic multi-slope noise RIRs -> reference-style batch dict."""
    rng = np.random.RandomState(seed)
    K = nfft // 2 + 1
    T60 = np.linspace(0.3, 1.5, G) if T60 is None else np.asarray(T60)
    t = np.arange(T)
    amps = rng.uniform(0.1, 1.0, (B, G))
    noise = rng.randn(B, T)
    env = np.exp(-6.908 * t[None, None, :] / (fs * T60[None, :, None]))
    rirs = (amps[:, :, None] * env).sum(1) * noise
    mix = int(20e-3 * fs)
    win = int(5e-3 * fs)
    full = np.fft.rfft(rirs, n=nfft, axis=-1)
    w = np.hanning(win)
    early = rirs[:, :mix].copy()
    late = rirs[:, mix:].copy()
    early[:, -(win // 2):] *= w[win // 2:]
    late[:, :win // 2] *= w[:win // 2]
    pos = rng.uniform(0, 1, (B, 3))
    z = np.exp(1j * 2 * np.pi * np.fft.rfftfreq(nfft))
    return {
        'z_values': torch.tensor(z),
        'source_position': torch.zeros(B, 3, dtype=torch.float64),
        'listener_position': torch.tensor(pos * 10.0),
        'norm_listener_position': torch.tensor(pos),
        'target_early_response': torch.tensor(np.fft.rfft(early, n=nfft, axis=-1)),
        'target_late_response': torch.tensor(np.fft.rfft(late, n=nfft, axis=-1)),
        'target_rir_response': torch.tensor(full),
    }, T60


def batch_to_np(batch, prefix='batch_'):
    return {prefix + k: c2np(v) for k, v in batch.items()}


def make_grid_model(fs, G, nper, delays, T60, seed, use_zero_coupling=True, layers=2, neurons=16,
                    nff=4):
    torch.manual_seed(seed)
    np.random.seed(seed)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR,
                            use_zero_coupling=use_zero_coupling)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=layers,
                            num_neurons_per_layer=neurons, num_fourier_features=nff)
    net = DiffGFDNVarReceiverPos(fs, G, delays, 'cpu', fl, of, use_absorption_filters=False,
                                 common_decay_times=np.asarray(T60)[None, :],
                                 use_colorless_loss=True)
    return net


def state_np(net, prefix='sd_'):
    return {prefix + k: c2np(v) for k, v in net.state_dict().items()}


# ----------------------------------------------------------------------------------------------
def gen_f1_feedback_loop():
    """F1: FeedbackLoop -> A, P, grads of sum|P|^2 (zero coupling and learnable coupling)."""
    out = {}
    fs = 8000.0
    nfft = 512
    z = torch.tensor(np.exp(1j * 2 * np.pi * np.fft.rfftfreq(nfft)))
    for tag, G, nper, zero in (('zc', 3, 4, True), ('cp', 3, 4, False), ('g1', 1, 6, True)):
        torch.manual_seed(11)
        delays = prime_delays(G * nper, lo=20, hi=100, seed=1)
        T60 = np.linspace(0.1, 0.4, G)
        from diff_gfdn.absorption_filters import decay_times_to_gain_per_sample
        gains = torch.flatten(torch.tensor([
            decay_times_to_gain_per_sample(T60[i], delays[i * nper:(i + 1) * nper], fs).tolist()
            for i in range(G)]))
        loop = FeedbackLoop(fs, G, nper, torch.tensor(delays, dtype=torch.float32), False,
                            coupling_matrix_type=CouplingMatrixType.SCALAR,
                            use_zero_coupling=zero, gains=gains)
        P = loop(z)
        loss = (P.abs() ** 2).sum()
        loss.backward()
        out[f'{tag}_delays'] = np.array(delays)
        out[f'{tag}_gamma'] = c2np(gains)
        out[f'{tag}_M'] = c2np(loop.M)
        out[f'{tag}_alpha'] = c2np(loop.alpha)
        out[f'{tag}_A'] = c2np(loop.coupled_feedback_matrix.real)
        out[f'{tag}_P'] = c2np(P)
        out[f'{tag}_grad_M'] = c2np(loop.M.grad)
        if not zero:
            out[f'{tag}_grad_alpha'] = c2np(loop.alpha.grad)
    out['z'] = c2np(z)
    out['fs'] = fs
    np.savez_compressed(os.path.join(HERE, 'f1_feedback_loop.npz'), **out)
    print('F1 done')


def gen_f2_f3_f4(tag, G, nper, nfft, fs, B, T, win, hop, zero=True, seed=5, use_asym=True):
    """F2 model forward, F3 losses + grads wrt H, F4 one normalize + train_step with Adam."""
    delays = prime_delays(G * nper, lo=int(20e-3 * fs), hi=int(50e-3 * fs), seed=seed)
    batch, T60 = synth_batch(B, nfft, fs, G, T, seed)
    net = make_grid_model(fs, G, nper, delays, T60, seed, use_zero_coupling=zero)
    out = {'fs': fs, 'nfft': nfft, 'G': G, 'nper': nper, 'delays': np.array(delays),
           'T60': T60, 'win': win, 'hop': hop, 'zero_coupling': zero}
    out.update(batch_to_np(batch))
    out.update(state_np(net))

    # ---- F2 forward
    H, (Hout, Hpd) = net(batch)
    out['H'] = c2np(H)
    out['Hout'] = c2np(Hout)
    out['Hout_per_del_nz'] = np.stack([c2np(Hpd[g * nper:(g + 1) * nper, :, g]) for g in range(G)])
    out['receiver_gains'] = c2np(net.output_scalars.gains)

    # ---- F3 losses on (target, H) with grads wrt H
    Hd = H.detach().clone().requires_grad_(True)
    tgt = batch['target_rir_response']
    e1 = edr_loss(fs, win_size=win, hop_size=hop)
    l_edr = e1(tgt, Hd)
    g_edr, = torch.autograd.grad(l_edr, Hd)
    max_ms = float(np.max(T60)) * 1e3
    e2 = edc_loss(max_ms, fs, use_mask=False)
    l_edc = e2(tgt, Hd)
    g_edc, = torch.autograd.grad(l_edc, Hd)
    out['loss_edr'] = l_edr.item()
    out['loss_edc'] = l_edc.item()
    out['grad_edr_H'] = c2np(g_edr)
    out['grad_edc_H'] = c2np(g_edc)
    Ho = Hout.detach().clone().requires_grad_(True)
    for nm, crit in (('mse', mse_loss()), ('amse', amse_loss())):
        vals, grads = [], []
        for k in range(G):
            l = crit(Ho[..., k], torch.ones_like(Ho[..., k]))
            g, = torch.autograd.grad(l, Ho)
            vals.append(l.item())
            grads.append(c2np(g[..., k]))
        out[f'loss_{nm}'] = np.array(vals)
        out[f'grad_{nm}_Hout'] = np.stack(grads)
    out['loss_sparsity'] = np.array([
        sparsity_loss()(net.feedback_loop.ortho_param(net.feedback_loop.M[k])).item()
        for k in range(G)])

    # ---- F4 full step: normalize + train_step (Adam), reference trainer
    import tempfile
    tmp = tempfile.mkdtemp()
    tc = TrainerConfig(batch_size=B, num_freq_bins=nfft, max_epochs=1, lr=1e-3, io_lr=1e-2,
                       coupling_angle_lr=1e-2, use_colorless_loss=True,
                       use_asym_spectral_loss=use_asym, edc_loss_weight=10.0,
                       edr_loss_weight=1.0, spectral_loss_weight=1.0, sparsity_loss_weight=2.0,
                       use_edc_mask=False, train_dir=tmp + '/t', ir_dir=tmp + '/a')
    trainer = VarReceiverPosTrainer(net, tc)
    # scaled-down STFT for the small cases (the reference hard-codes 4096/2048 defaults)
    trainer.criterion[0].win_size = win
    trainer.criterion[0].hop_size = hop
    trainer.normalize(batch)
    out.update(state_np(net, 'sdn_'))          # after normalize
    net.zero_grad()
    Hn, Hs = net(batch)
    all_losses = trainer.calculate_losses(batch, Hn, Hs)
    total = sum(all_losses.values())
    total.backward()
    for k, v in all_losses.items():
        out['step_' + k] = float(v.item())
    out['step_total'] = float(total.item())
    for name, prm in net.named_parameters():
        out['grad_' + name] = c2np(prm.grad)
    trainer.optimizer.step()
    out.update(state_np(net, 'sda_'))          # after Adam
    np.savez_compressed(os.path.join(HERE, f'f234_{tag}.npz'), **out)
    print('F2-4', tag, 'done', {k: round(float(v), 6) for k, v in out.items() if k.startswith('step_')})


def gen_f3_subband_mask():
    """F3b: sub-band filter multiply + masked EDC + frequency-weighted EDR on given (target, H)."""
    fs, nfft, B, G, nper = 8000.0, 2048, 3, 2, 3
    batch, T60 = synth_batch(B, nfft, fs, G, 1500, 77)
    torch.manual_seed(3)
    K = nfft // 2 + 1
    H = (torch.randn(B, K, dtype=torch.float64) + 1j * torch.randn(B, K, dtype=torch.float64)) * 0.05 \
        + batch['target_rir_response'] * 0.8
    taps = np.random.RandomState(4).randn(65) * np.hanning(65)
    filt = torch.fft.rfft(torch.tensor(taps), n=nfft)
    Hd = H.clone().requires_grad_(True)
    Hs = Hd * filt
    tgt = batch['target_rir_response']
    out = {'fs': fs, 'nfft': nfft, 'H': c2np(H), 'filt': c2np(filt), 'target': c2np(tgt),
           'T60': T60}
    e1 = edr_loss(fs, win_size=256, hop_size=128, use_weight_fn=True)
    l1 = e1(tgt, Hs)
    g1, = torch.autograd.grad(l1, Hd, retain_graph=True)
    out['freq_weights'] = c2np(e1.frequency_weights)
    out['loss_edr_w'] = l1.item()
    out['grad_edr_w'] = c2np(g1)
    # masked EDC: capture the mask the reference draws
    max_ms = float(np.max(T60)) * 1e3
    e2 = edc_loss(max_ms, fs, use_mask=True)
    torch.manual_seed(99)
    L = min(e2.max_ir_len_samps, K) - e2.mixing_time_samps
    probs = torch.empty(L).uniform_(0, 1)
    mask_index = torch.argwhere(torch.bernoulli(probs))
    torch.manual_seed(99)
    l2 = e2(tgt, Hs)
    g2, = torch.autograd.grad(l2, Hd)
    out['edc_mask_index'] = c2np(mask_index)
    out['loss_edc_masked'] = l2.item()
    out['grad_edc_masked'] = c2np(g2)
    np.savez_compressed(os.path.join(HERE, 'f3b_subband_mask.npz'), **out)
    print('F3b done', out['loss_edr_w'], out['loss_edc_masked'])


def gen_f5_single_pos():
    fs, nfft, G, nper = 8000.0, 1024, 2, 4
    delays = prime_delays(G * nper, lo=160, hi=400, seed=2)
    batch, T60 = synth_batch(1, nfft, fs, G, 900, 21)
    torch.manual_seed(8)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=False)
    of = OutputFilterConfig(use_svfs=False)
    net = DiffGFDNSinglePos(fs, G, delays, 'cpu', fl, of, use_absorption_filters=False,
                            common_decay_times=np.asarray(T60)[None, :], use_colorless_loss=True)
    x = {'z_values': batch['z_values'],
         'target_early_response': batch['target_early_response'][0].clone(),
         'target_rir_response': batch['target_rir_response'][0],
         'target_late_response': batch['target_late_response'][0]}
    H, (Hout, Hpd) = net(x)
    out = {'fs': fs, 'nfft': nfft, 'G': G, 'nper': nper, 'delays': np.array(delays), 'T60': T60,
           'H': c2np(H), 'Hout': c2np(Hout),
           'z': c2np(batch['z_values']), 'early': c2np(batch['target_early_response'][0]),
           'target': c2np(batch['target_rir_response'][0])}
    out.update(state_np(net))
    # single-item losses (2-D EDR branch, losses.py:493-495)
    Hd = H.detach().clone().requires_grad_(True)
    l = edr_loss(fs, win_size=256, hop_size=128)(x['target_rir_response'], Hd)
    g, = torch.autograd.grad(l, Hd)
    out['loss_edr'] = l.item()
    out['grad_edr_H'] = c2np(g)
    np.savez_compressed(os.path.join(HERE, 'f5_single_pos.npz'), **out)
    print('F5 done')


def gen_f6_directional():
    """F6: directional model with a patched analysis matrix and decay-kernel envelopes."""
    fs, nfft, G, order, B, J = 8000.0, 1024, 2, 1, 3, 5
    nper = (order + 1) ** 2
    delays = prime_delays(G * nper, lo=160, hi=400, seed=4)
    batch, T60 = synth_batch(B, nfft, fs, G, 900, 31)
    rng = np.random.RandomState(12)
    analysis = rng.randn(J, nper).astype(np.float32)

    def fake_init(self, beamformer_type, desired_directions):
        self.modal_weights = np.ones(self.ambi_order + 1)
        self.analysis_matrix = torch.tensor(analysis, dtype=torch.float32)

    ss_model.Directional_Beamforming_Weights.initialise_beamformer = fake_init

    def fake_decay_kernel(t_vals, time_axis, fs_, normalize_envelope=True, add_noise=False):
        # stated formula (build side): exp(-13.8155 t / T60), NOT the slope2noise source
        return np.exp(-13.815510557964274 * time_axis[:, None] / np.asarray(t_vals).reshape(1, -1))

    ref_losses.decay_kernel = fake_decay_kernel
    torch.manual_seed(13)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=1, num_neurons_per_layer=8,
                            num_fourier_features=3, use_skip_connections=False)
    cdt = np.asarray(T60)[None, :]
    net = DiffDirectionalFDNVarReceiverPos(fs, G, delays, 'cpu', fl, of, ambi_order=order,
                                           desired_directions=np.zeros((2, J)),
                                           common_decay_times=cdt, use_colorless_loss=True)
    H_sh, (Hout, _) = net(batch)
    sh_gains = net.sh_output_scalars.weights
    H_dir = torch.einsum('jl, blk -> bjk',
                         torch.complex(net.sh_output_scalars.analysis_matrix,
                                       torch.zeros_like(net.sh_output_scalars.analysis_matrix)), H_sh)
    edc_len_ms = float(np.max(T60)) * 1e3 * 0.1
    crit = directional_edc_loss(cdt, edc_len_ms, fs, use_mask=False)
    amps = torch.tensor(rng.uniform(0.1, 1.0, (B, J, G)))
    loss = crit(H_dir, amps)
    loss.backward()
    out = {'fs': fs, 'nfft': nfft, 'G': G, 'nper': nper, 'order': order, 'delays': np.array(delays),
           'T60': T60, 'analysis_matrix': analysis, 'envelopes': c2np(crit.envelopes),
           'edc_len_ms': edc_len_ms, 'amps': c2np(amps), 'H_sh': c2np(H_sh), 'H_dir': c2np(H_dir),
           'sh_gains': c2np(sh_gains), 'Hout': c2np(Hout), 'loss': loss.item()}
    out.update(batch_to_np(batch))
    out.update(state_np(net))
    for name, prm in net.named_parameters():
        if prm.grad is not None:
            out['grad_' + name] = c2np(prm.grad)
    np.savez_compressed(os.path.join(HERE, 'f6_directional.npz'), **out)
    print('F6 done', loss.item())


def gen_f7_front_end():
    """F7: RoomDataset front end (rfft x3, in-place fade quirk, z grid)."""
    from diff_gfdn.dataloader import RoomDataset, MultiRIRDataset, custom_collate
    rng = np.random.RandomState(41)
    fs, nfft, R, T = 8000.0, 1024, 4, 700
    rirs = rng.randn(R, T) * np.exp(-np.arange(T) / 150.0)[None, :]
    pos = rng.uniform(0, 5, (R, 3))
    src = np.array([1.0, 2.0, 1.5])

    class _DS(RoomDataset):
        def get_2D_meshgrid(self):
            return None

    ds = _DS(2, fs, src, pos, rirs.copy(), np.array([[0.2, 0.4]]), [[1, 1, 1]] * 2,
             [[0, 0, 0]] * 2, nfft=nfft)
    md = MultiRIRDataset('cpu', ds)
    batch = custom_collate([md[i] for i in range(R)])
    out = {'fs': fs, 'nfft': nfft, 'rirs': rirs, 'pos': pos, 'src': src}
    out.update(batch_to_np(batch))
    out['rirs_after'] = ds.rirs
    np.savez_compressed(os.path.join(HERE, 'f7_front_end.npz'), **out)
    print('F7 done')


def gen_f8_source_receiver():
    """DiffGFDNVarSourceReceiverPos (model.py:303-452): gains from the receiver AND the source position."""
    fs, nfft, G, nper, B = 2000.0, 512, 3, 4, 4
    delays = prime_delays(G * nper, lo=20, hi=90, seed=4)
    for tag, zero in (('zc', True), ('cp', False)):
        batch, T60 = synth_batch(B, nfft, fs, G, 400, 31)
        rng = np.random.RandomState(77)
        batch['source_position'] = torch.tensor(rng.uniform(0, 1, (B, 3)))
        torch.manual_seed(12)
        np.random.seed(12)
        fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=zero)
        of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
        net = DiffGFDNVarSourceReceiverPos(fs, G, delays, 'cpu', fl, of, of, use_absorption_filters=False,
                                           learn_common_decay_times=False,
                                           common_decay_times=np.asarray(T60)[None, :], use_colorless_loss=True)
        H, (Hout, _) = net(batch)
        loss = (H.abs() ** 2).sum()
        loss.backward()
        out = {'fs': fs, 'nfft': nfft, 'G': G, 'nper': nper, 'delays': np.array(delays), 'T60': T60,
               'zero_coupling': zero, 'H': c2np(H), 'Hout': c2np(Hout), 'loss': loss.item()}
        out.update(batch_to_np(batch))
        out.update(state_np(net))
        out.update({'grad_' + k: c2np(p.grad) for k, p in net.named_parameters() if p.grad is not None})
        np.savez_compressed(os.path.join(HERE, f'f8_source_receiver_{tag}.npz'), **out)
    print('F8 done')


def gen_f16_source_receiver_svf():
    """DiffGFDNVarSourceReceiverPos with SVF filters from MLPs on the input (source position) and the output (receiver
    position) side (model.py:305-400 constructor branches, :402-452 forward; gain_filters.py:262-402), zero coupling and
    learnable coupling; plus the mixed case (SVF output filters, scalar input gains)."""
    fs, nfft, G, nper, B = 8000.0, 512, 2, 4, 3
    delays = prime_delays(G * nper, lo=160, hi=400, seed=8)
    for tag, zero, svf_in in (('zc', True, True), ('cp', False, True), ('mixed', True, False)):
        batch, T60 = synth_batch(B, nfft, fs, G, 450, 61, T60=[0.3, 0.6])
        rng = np.random.RandomState(78)
        batch['source_position'] = torch.tensor(rng.uniform(0, 1, (B, 3)))
        torch.manual_seed(33)
        np.random.seed(33)
        fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=zero)
        of = OutputFilterConfig(use_svfs=True, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4,
                                compress_pole_factor=0.98)
        inf = OutputFilterConfig(use_svfs=svf_in, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4,
                                 compress_pole_factor=0.98)
        net = DiffGFDNVarSourceReceiverPos(fs, G, delays, 'cpu', fl, of, inf, use_absorption_filters=False,
                                           learn_common_decay_times=False,
                                           common_decay_times=np.asarray(T60)[None, :], use_colorless_loss=True)
        H, (Hout, _) = net(batch)
        loss = (H.abs() ** 2).sum()
        loss.backward()
        out = {'fs': fs, 'nfft': nfft, 'G': G, 'nper': nper, 'delays': np.array(delays), 'T60': T60,
               'zero_coupling': zero, 'svf_in': svf_in, 'H': c2np(H), 'Hout': c2np(Hout), 'loss': loss.item(),
               'Co': c2np(net.output_filters(batch)[:, ::nper, :])}
        if svf_in:
            out['Ci'] = c2np(net.input_filters(batch)[:, ::nper, :])
        out.update(batch_to_np(batch))
        out.update(state_np(net))
        out.update({'grad_' + k: c2np(p.grad) for k, p in net.named_parameters() if p.grad is not None})
        np.savez_compressed(os.path.join(HERE, f'f16_source_receiver_svf_{tag}.npz'), **out)
    print('F16 done')


def gen_f9_colorless_fdn():
    """ColorlessFDN prototype (colorless_fdn/model.py:12-111) + the trainer's loss and normalisation
    (colorless_fdn/trainer.py:95-143)."""
    from diff_gfdn.colorless_fdn.model import ColorlessFDN
    from diff_gfdn.colorless_fdn.losses import amse_loss as c_amse, sparsity_loss as c_sparse
    from diff_gfdn.utils import get_frequency_samples
    fs, N = 8000.0, 8
    delays = prime_delays(N, lo=160, hi=400, seed=6)
    torch.manual_seed(15)
    np.random.seed(15)
    net = ColorlessFDN(fs, delays, 'cpu', nominal_t60=10.0)
    z = get_frequency_samples(600)
    out = {'fs': fs, 'delays': np.array(delays), 'z': c2np(z)}
    out.update(state_np(net))
    H, Hpd = net(z)
    out['H'], out['Hpd'] = c2np(H), c2np(Hpd)
    fl = net.feedback_loop
    loss = c_amse()(H, torch.ones(len(z))) + 1.5 * c_sparse()(fl.ortho_param(fl.random_feedback_matrix))
    loss.backward()
    out['loss'] = loss.item()
    out.update({'grad_' + k: c2np(p.grad) for k, p in net.named_parameters()})
    vloss = c_amse()(H, torch.ones(len(z))) + c_amse()(Hpd, torch.ones_like(Hpd)) \
        + 1.5 * c_sparse()(fl.ortho_param(fl.random_feedback_matrix))
    out['valid_loss'] = vloss.item()
    with torch.no_grad():                                   # trainer.normalize (:133-143)
        energy = torch.sum(torch.abs(H) ** 2) / H.shape[0]
        out['norm_input_gains'] = c2np(net.input_gains / torch.pow(energy, 1 / 4))
        out['norm_output_gains'] = c2np(net.output_gains / torch.pow(energy, 1 / 4))
    np.savez_compressed(os.path.join(HERE, 'f9_colorless_fdn.npz'), **out)
    print('F9 done')


def gen_f10_absorption_filters():
    """DiffGFDNVarReceiverPos with frequency-dependent absorption (use_absorption_filters: GEQ-designed SOS per
    delay line, model.py:131-153, feedback_loop.py:332-344): the designed coefficients are stored as data."""
    fs, nfft, G, nper, B = 8000.0, 1024, 2, 4, 3
    delays = prime_delays(G * nper, lo=160, hi=400, seed=3)
    band_centre_hz = [125.0, 250.0, 500.0, 1000.0, 2000.0]
    T60 = np.stack([np.linspace(0.5, 0.25, 5), np.linspace(0.9, 0.4, 5)], axis=1)     # (bands, G)
    batch, _ = synth_batch(B, nfft, fs, G, 900, 41, T60=[0.4, 0.7])
    torch.manual_seed(19)
    np.random.seed(19)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=False)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    net = DiffGFDNVarReceiverPos(fs, G, delays, 'cpu', fl, of, use_absorption_filters=True,
                                 common_decay_times=T60, band_centre_hz=band_centre_hz, use_colorless_loss=True)
    H, (Hout, _) = net(batch)
    loss = (H.abs() ** 2).sum()
    loss.backward()
    out = {'fs': fs, 'nfft': nfft, 'G': G, 'nper': nper, 'delays': np.array(delays), 'T60': T60,
           'band_centre_hz': np.array(band_centre_hz), 'H': c2np(H), 'Hout': c2np(Hout), 'loss': loss.item(),
           'P_small': c2np(net.feedback_loop(batch['z_values'][:64]))}
    out.update(batch_to_np(batch))
    out.update(state_np(net))
    out.update({'grad_' + k: c2np(p.grad) for k, p in net.named_parameters() if p.grad is not None})
    np.savez_compressed(os.path.join(HERE, 'f10_absorption_filters.npz'), **out)
    print('F10 done', out['sd_delay_filters'].shape)


def gen_f11_svf_filters():
    """SVF output filters from an MLP on the grid model (gain_filters.py:262-402, model.py:544-592) and learnable
    SVF input + output filters on the single-position model (model.py:723-778, get_filter :838-911)."""
    fs, nfft, G, nper, B = 8000.0, 512, 2, 4, 3
    delays = prime_delays(G * nper, lo=160, hi=400, seed=8)
    batch, T60 = synth_batch(B, nfft, fs, G, 450, 51, T60=[0.3, 0.6])
    out = {'fs': fs, 'nfft': nfft, 'G': G, 'nper': nper, 'delays': np.array(delays), 'T60': T60}
    out.update(batch_to_np(batch))
    # (a) grid model, zero coupling
    torch.manual_seed(23)
    np.random.seed(23)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=True, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4,
                            compress_pole_factor=0.98)
    net = DiffGFDNVarReceiverPos(fs, G, delays, 'cpu', fl, of, use_absorption_filters=False,
                                 common_decay_times=np.asarray(T60)[None, :], use_colorless_loss=True)
    H, (Hout, _) = net(batch)
    loss = (H.abs() ** 2).sum()
    loss.backward()
    out.update({'grid_H': c2np(H), 'grid_Hout': c2np(Hout), 'grid_Co': c2np(net.output_filters(batch)[:, ::nper, :])})
    out.update(state_np(net, 'gsd_'))
    out.update({'ggrad_' + k: c2np(p.grad) for k, p in net.named_parameters() if p.grad is not None})
    # (b) single position, coupled, SVFs on both sides
    torch.manual_seed(29)
    np.random.seed(29)
    fl2 = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=False)
    of2 = OutputFilterConfig(use_svfs=True, compress_pole_factor=1.0)
    sp = DiffGFDNSinglePos(fs, G, delays, 'cpu', fl2, of2, use_absorption_filters=False,
                           common_decay_times=np.asarray(T60)[None, :], use_colorless_loss=True,
                           input_filter_config=of2)
    x = {'z_values': batch['z_values'], 'target_early_response': batch['target_early_response'][0].clone()}
    # (the reference's get_filter deep-copies non-leaf tensors, model.py:905-908, which torch refuses while
    # recording gradients: its SVF single-position forward only runs under no_grad -- forward values are pinned)
    with torch.no_grad():
        Hs, (Hsout, _) = sp(x)
    out.update({'sp_H': c2np(Hs), 'sp_Hout': c2np(Hsout)})
    out.update(state_np(sp, 'ssd_'))
    np.savez_compressed(os.path.join(HERE, 'f11_svf_filters.npz'), **out)
    print('F11 done')


def full_convolve(a, b, mode='full'):
    """Stand-in for torchaudio.functional.convolve(mode='full') on 1-D tensors (torchaudio is not in this image;
    SURVEY §8c(iv)): the full linear convolution, through conv1d so that autograd follows it."""
    assert mode == 'full' and a.ndim == 1 and b.ndim == 1
    return torch.nn.functional.conv1d(a[None, None], b.flip(0)[None, None], padding=len(b) - 1).reshape(-1)


def gen_f12_filter_coupling():
    """Paraunitary FILTER coupling (feedback_loop.py:90-143, :311-323, :362-373, :413-455): the loop alone
    (Phi, A, P and the gradients of sum |P|^2) and the grid model on top of it."""
    import types
    import diff_gfdn.utils as ref_utils
    ref_utils.Faudio = types.SimpleNamespace(convolve=full_convolve)
    from diff_gfdn.absorption_filters import decay_times_to_gain_per_sample
    out = {}
    fs, nfft, G, nper, order = 8000.0, 512, 3, 4, 6
    z = torch.tensor(np.exp(1j * 2 * np.pi * np.fft.rfftfreq(nfft)))
    torch.manual_seed(31)
    delays = prime_delays(G * nper, lo=20, hi=100, seed=2)
    T60 = np.linspace(0.1, 0.4, G)
    gains = torch.flatten(torch.tensor([
        decay_times_to_gain_per_sample(T60[i], delays[i * nper:(i + 1) * nper], fs).tolist() for i in range(G)]))
    loop = FeedbackLoop(fs, G, nper, torch.tensor(delays, dtype=torch.float32), False,
                        coupling_matrix_type=CouplingMatrixType.FILTER, coupling_matrix_order=order, gains=gains)
    P = loop(z)
    loss = (P.abs() ** 2).sum()
    loss.backward()
    out.update({'loop_delays': np.array(delays), 'loop_gamma': c2np(gains), 'loop_M': c2np(loop.M),
                'loop_unit_vectors': c2np(loop.unit_vectors), 'loop_unitary_matrix': c2np(loop.unitary_matrix),
                'loop_phi': c2np(loop.phi), 'loop_A': c2np(loop.coupled_feedback_matrix), 'loop_P': c2np(P),
                'loop_grad_M': c2np(loop.M.grad), 'loop_grad_unit_vectors': c2np(loop.unit_vectors.grad),
                'loop_grad_unitary_matrix': c2np(loop.unitary_matrix.grad), 'z': c2np(z), 'fs': fs,
                'order': order})
    # grid model
    G2, nper2, B, order2 = 2, 4, 3, 5
    delays2 = prime_delays(G2 * nper2, lo=160, hi=400, seed=6)
    batch, T60b = synth_batch(B, 1024, fs, G2, 900, 61, T60=[0.4, 0.7])
    torch.manual_seed(37)
    np.random.seed(37)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.FILTER, pu_matrix_order=order2)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    net = DiffGFDNVarReceiverPos(fs, G2, delays2, 'cpu', fl, of, use_absorption_filters=False,
                                 common_decay_times=np.asarray(T60b)[None, :], use_colorless_loss=True)
    H, (Hout, _) = net(batch)
    lossm = (H.abs() ** 2).sum()
    lossm.backward()
    out.update({'net_nfft': 1024, 'net_G': G2, 'net_nper': nper2, 'net_order': order2, 'net_delays': np.array(delays2),
                'net_T60': T60b, 'net_H': c2np(H), 'net_Hout': c2np(Hout), 'net_loss': lossm.item()})
    out.update(batch_to_np(batch, 'net_batch_'))
    out.update(state_np(net, 'net_sd_'))
    out.update({'net_grad_' + k: c2np(p.grad) for k, p in net.named_parameters() if p.grad is not None})
    np.savez_compressed(os.path.join(HERE, 'f12_filter_coupling.npz'), **out)
    print('F12 done', sorted(k for k in out if k.startswith('net_grad_')))


def gen_f15_full_band_and_filter_absorption():
    """(a) The full-band configuration as the reference runs it (data/config/treble_data_grid_training_full_band_
    colorless_loss.yml:5-6, :22-26): SVF output filters from a 10 x 64 MLP on 20 Fourier features TOGETHER WITH
    absorption filters on the delay lines, N = 12 = 3 groups x 4, scalar coupling (gain_filters.py:262-402,
    feedback_loop.py:332-344, :376-381, model.py:544-619).  (b) FILTER coupling together with absorption filters: the
    reference's FeedbackLoop.forward handles both in one pass (feedback_loop.py:362-386)."""
    import types
    import diff_gfdn.utils as ref_utils
    ref_utils.Faudio = types.SimpleNamespace(convolve=full_convolve)
    fs, nfft, G, nper, B = 8000.0, 1024, 3, 4, 3
    band_centre_hz = [125.0, 250.0, 500.0, 1000.0, 2000.0]
    T60 = np.stack([np.linspace(0.5, 0.25, 5), np.linspace(0.9, 0.4, 5), np.linspace(0.7, 0.3, 5)], axis=1)   # (bands, G)
    delays = prime_delays(G * nper, lo=160, hi=400, seed=12)
    batch, _ = synth_batch(B, nfft, fs, G, 900, 71, T60=[0.4, 0.7, 0.55])
    out = {'fs': fs, 'nfft': nfft, 'G': G, 'nper': nper, 'delays': np.array(delays), 'T60': T60,
           'band_centre_hz': np.array(band_centre_hz)}
    out.update(batch_to_np(batch))
    # (a) SVF output filters + absorption filters, the YAML's network
    torch.manual_seed(43)
    np.random.seed(43)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR)
    of = OutputFilterConfig(use_svfs=True, num_hidden_layers=10, num_neurons_per_layer=64, num_fourier_features=20)
    net = DiffGFDNVarReceiverPos(fs, G, delays, 'cpu', fl, of, use_absorption_filters=True, common_decay_times=T60,
                                 band_centre_hz=band_centre_hz, use_colorless_loss=True)
    H, (Hout, _) = net(batch)
    loss = (H.abs() ** 2).sum()
    loss.backward()
    out.update({'fb_H': c2np(H), 'fb_Hout': c2np(Hout), 'fb_loss': loss.item(),
                'fb_compress_pole_factor': float(of.compress_pole_factor),
                'fb_use_zero_coupling': bool(fl.use_zero_coupling)})
    out.update(state_np(net, 'fb_sd_'))
    out.update({'fb_grad_' + k: c2np(p.grad) for k, p in net.named_parameters() if p.grad is not None})
    # (b) FILTER coupling + absorption filters
    order = 4
    torch.manual_seed(47)
    np.random.seed(47)
    fl2 = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.FILTER, pu_matrix_order=order)
    of2 = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    net2 = DiffGFDNVarReceiverPos(fs, G, delays, 'cpu', fl2, of2, use_absorption_filters=True, common_decay_times=T60,
                                  band_centre_hz=band_centre_hz, use_colorless_loss=True)
    H2, (Hout2, _) = net2(batch)
    loss2 = (H2.abs() ** 2).sum()
    loss2.backward()
    out.update({'fa_order': order, 'fa_H': c2np(H2), 'fa_Hout': c2np(Hout2), 'fa_loss': loss2.item(),
                'fa_P_small': c2np(net2.feedback_loop(batch['z_values'][:48]))})
    out.update(state_np(net2, 'fa_sd_'))
    out.update({'fa_grad_' + k: c2np(p.grad) for k, p in net2.named_parameters() if p.grad is not None})
    np.savez_compressed(os.path.join(HERE, 'f15_full_band.npz'), **out)
    print('F15 done', out['fb_sd_delay_filters'].shape, sorted(k for k in out if k.startswith('fa_grad_')))


def gen_f13_single_rir_data():
    """RIRData / SingleRIRDataset (dataloader.py:76-180, :603-658): responses of one RIR and the z grid."""
    from diff_gfdn.dataloader import RIRData, SingleRIRDataset
    rng = np.random.RandomState(13)
    fs = 8000.0
    rir = rng.randn(6000) * np.exp(-np.arange(6000) / 1500.0)
    out = {'fs': fs, 'rir': rir.copy(), 'T60': np.array([[0.6]])}
    d = RIRData(np.array([[0.6]]), None, rir=rir.copy(), sample_rate=fs, nfft=8192, mixing_time_ms=20.0)
    ds = SingleRIRDataset('cpu', d, new_sampling_radius=1.0002)
    out.update({'nfft': 8192, 'rir_after': d.rir.copy(), 'full': d.rir_mag_response, 'early': d.early_rir_mag_response,
                'late': d.late_rir_mag_response, 'z': c2np(ds.z_values), 'auto_bins': RIRData(
                    np.array([[0.6]]), None, rir=rir.copy(), sample_rate=fs).num_freq_bins})
    np.savez_compressed(os.path.join(HERE, 'f13_single_rir_data.npz'), **out)
    print('F13 done')


def gen_f14_learnable_decay_times():
    """F14: learnable common decay times (feedback_loop.py:205-232, model.py:127-153): the gains are the
    differentiable function 10^(-3 m / (fs T60_g)) of an nn.Parameter; forward H, decay losses, gradients of every
    parameter incl. feedback_loop.common_decay_times.  (The reference evaluates the gains ONCE, in the constructor:
    the fixture is the first forward / backward, where that equals re-evaluating them.)"""
    fs, nfft, G, nper, B, T, win, hop, seed = 4000.0, 1024, 3, 4, 3, 900, 128, 64, 21
    delays = prime_delays(G * nper, lo=int(20e-3 * fs), hi=int(50e-3 * fs), seed=seed)
    batch, T60 = synth_batch(B, nfft, fs, G, T, seed)
    torch.manual_seed(seed)
    np.random.seed(seed)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    net = DiffGFDNVarReceiverPos(fs, G, delays, 'cpu', fl, of, use_absorption_filters=False,
                                 learn_common_decay_times=True, common_decay_times=np.asarray(T60)[None, :],
                                 use_colorless_loss=False)
    out = {'fs': fs, 'nfft': nfft, 'G': G, 'nper': nper, 'delays': np.array(delays), 'T60': T60, 'win': win, 'hop': hop}
    out.update(batch_to_np(batch))
    out.update(state_np(net))
    H = net(batch)
    out['H'] = c2np(H)
    tgt = batch['target_rir_response']
    l_edr = edr_loss(fs, win_size=win, hop_size=hop)(tgt, H)
    l_edc = edc_loss(float(np.max(T60)) * 1e3, fs, use_mask=False)(tgt, H)
    (l_edr + 10.0 * l_edc).backward()
    out['loss_edr'], out['loss_edc'] = l_edr.item(), l_edc.item()
    for name, prm in net.named_parameters():
        out['grad_' + name] = c2np(prm.grad)
    np.savez_compressed(os.path.join(HERE, 'f14_learnable_decay_times.npz'), **out)
    print('F14 done', l_edr.item(), l_edc.item(), out['grad_feedback_loop.common_decay_times'])


if __name__ == '__main__':
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == 'new':          # only the fixtures added in round 2
        gen_f2_f3_f4('n32_k1025', G=4, nper=8, nfft=2048, fs=8000.0, B=3, T=2000, win=256, hop=128, seed=13)
        gen_f14_learnable_decay_times()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'r3':           # only the fixture added in round 3
        gen_f15_full_band_and_filter_absorption()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'r4':           # only the fixture added in round 4
        gen_f16_source_receiver_svf()
        sys.exit(0)
    gen_f1_feedback_loop()
    # small: nfft 512 (K = 257, Fermat prime -> prime-length irfft quirk), scaled STFT
    gen_f2_f3_f4('n12_k257', G=3, nper=4, nfft=512, fs=2000.0, B=4, T=400, win=64, hop=32)
    # medium: nfft 8192 (K = 4097 = 17*241, composite), N = 16, learnable coupling
    gen_f2_f3_f4('n16_k4097_cp', G=4, nper=4, nfft=8192, fs=8000.0, B=3, T=5000, win=512,
                 hop=256, zero=False, seed=9, use_asym=False)
    gen_f3_subband_mask()
    gen_f5_single_pos()
    gen_f6_directional()
    gen_f7_front_end()
    gen_f8_source_receiver()
    gen_f9_colorless_fdn()
    gen_f10_absorption_filters()
    gen_f11_svf_filters()
    gen_f12_filter_coupling()
    gen_f13_single_rir_data()
    # N = 32 (4 groups x 8 lines: BASELINE config 5), nfft 2048 (K = 1025 = 5^2 41)
    gen_f2_f3_f4('n32_k1025', G=4, nper=8, nfft=2048, fs=8000.0, B=3, T=2000, win=256, hop=128, seed=13)
    gen_f14_learnable_decay_times()
    gen_f15_full_band_and_filter_absorption()
    gen_f16_source_receiver_svf()
