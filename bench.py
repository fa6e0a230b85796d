#!/usr/bin/env python
"""Benchmark of the DiffGFDN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling auto|weak|strong] [--epoch]

With N > 1 and no launcher around it (WORLD_SIZE unset) the script starts its own N ranks -- a child
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` -- BEFORE anything touches
the GPU, waits for it and exits with its code; under an outer torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.

Workload (the configuration BASELINE.json's metric is quoted on -- "N=16 delay lines, 838 pos x 7 bands" -- which fits
one GPU): 7 octave-band GFDNs (63 Hz ... 4 kHz), each N = 16 delay lines (4 groups x 4) on its own 838-receiver
synthetic grid, nfft = 131 072 (K = 65 537 bins), fs = 32 kHz, batch of 32 receiver positions per band per optimiser
step, losses EDR(w=1) + EDC(w=10, random mask) + asymmetric spectral(w=1) + sparsity(w=2), Adam -- the recipe of the
reference's src/run_subband_training_treble.py:105-154, which trains the bands one after another; here the band bank
(diffgfdn_amd/bandbank.py) steps all of them with one launch per stage.  ``--bands 1`` runs BASELINE.json configs[1]
alone (the 500 Hz band).

One "step" = what the reference's training loop does per batch (trainer.py:373-379), for every band: normalize
(no-grad sub-FDN forward + in-place rescale of b, c) + train_step (forward, losses, backward, [all-reduce], Adam).
Inputs are resident in HBM before the timed region.  Metric: RIR-frames/s = bands x receivers x 32 EDR frames /
second, summed over ranks.  ``--scaling auto`` (default): N = 1 the step at batch 32 ("weak"); N > 1 the headline is
STRONG scaling -- the reference's global batch of 32 per band split over the ranks -- with the weak figure (every rank
steps its own 32 receivers per band, global batch 32 N) and the band-sharded placement (whole bands over the ranks, no
collective) beside it in ``extra``; ``--scaling weak`` / ``strong``: that one figure only.  Either way the gradients of all
bands and the per-band loss terms are summed by ONE all-reduce of one flat buffer, between the step's two graphs (or inside
one graph with ``--captured-allreduce``).  At N = 1 the default line also carries the N = 32 configuration and the
directional configuration as short sub-records (``extra.n32``, ``extra.directional``).

The JSON line also carries
  roofline     : HBM roofline of the dominant kernel.  ``achieved`` uses its duration IN THE STEP: the same launch
                 sequence on the same streams as the timed graph (host launches), bracketed by HIP events on the
                 stream the kernel is launched on; the duration of the kernel alone on the chip is reported beside it
                 (``isolated_us``).  ``traffic`` is read from the committed PMC summary (profiles/), ``top`` lists the
                 longest kernels of the committed rocprofv3 stats with their algorithmic bytes, so that every
                 fraction can be recomputed from profiles/ alone;
  cpu_baseline : the same step on the host cores by the CPU oracle (oracle/cpu_trainer.py, a restatement pinned to
                 the reference; kind "port"), rank 0 at N = 1 only, with ``loss_delta_vs_cpu``: every loss term (the
                 masked EDC term included: the CPU step's mask is handed to the HIP step) and the gradients of the
                 timed HIP path against the CPU step on the same parameters and batch.
``--epoch``: times whole epochs of the reference's loop (trainer.py:345-424: 19 train steps incl. the ragged tail, 5
validation steps, StepLR, checkpoint of every band) instead of bare steps.
"""
import argparse
import csv
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 32000.0
NFFT = 131072
K = NFFT // 2 + 1
KU = (K + 1) // 2
NUM_RECEIVERS = 838
G, NPER = 4, 4
BATCH = 32
BAND_CENTRES = (63.0, 125.0, 250.0, 500.0, 1000.0, 2000.0, 4000.0)
FRAMES = 32
WIN = 4096
NF = WIN // 2 + 1
EDC_LEN = 48000 - 640
HBM_PEAK_GBS = 8000.0
# compulsory HBM bytes per RIR (fwd + bwd, fp32 / complex64 storage)
ALG_BYTES_PER_RIR = 2811048
# ... and what the LINEAR step (rounds 4-5) has to move per RIR: the receiver's transformed direct path as short-time spectra
# (32 x 2049 x 8 B) and on the EDC window (4 x 47 360 B), the target EDR (32 x 2049 x 4 B) and the target EDC (4 x 47 360 B);
# everything else of the step is per band (group signals, records), not per receiver
LINEAR_ALG_BYTES_PER_RIR = FRAMES * NF * (8 + 4) + 2 * 4 * EDC_LEN          # = 1 165 696
# Dominant HBM-bound kernel of the step since round 4: the EDR loss on composed short-time spectra with the sums over the
# band's receivers (csrc/edrlin.hip, k_edr_lin_wave), one launch per step.  Per RIR it MUST read the STFT of the receiver's
# transformed direct path (32 frames x 2049 bins x 8 B) and the target EDR (x 4 B) = 786 816 B; the band's group spectra in
# and the band's gradient spectra out (29 MB per launch together) are shared by the band's 32 receivers and NOT counted.
DOMINANT_KERNEL = 'k_edr_lin_wave'
# ... and beside it, on the side stream, the EDC term's register-resident launch (csrc/edcone.hip, k_edc_lin_one).  The two run
# beside each other and whichever starts second is stretched: the roofline leg brackets BOTH with HIP events, names the one
# that is longer in the step (``roofline.kernel``) and reports the pair's window and bytes (``roofline.pair``).
PAIR_KERNELS = ('k_edr_lin_wave', 'k_edc_lin_one')
# Algorithmic HBM bytes per RIR and launch of the step's per-receiver kernels (what each MUST read and write):
ALG_BYTES_PER_UNIT = {
    'k_edr_lin_wave': FRAMES * NF * (8 + 4),                                   # Sd, target EDR in (the G sums out: per band)
    'k_edr_lin_cols': FRAMES * NF * (8 + 4 + 4),                               # Sd, target EDR in; dL/d|S|^2 out
    'k_edr_lin_gsum': FRAMES * NF * (8 + 4),                                   # Sd, dL/d|S|^2 in (the G sums out: per band)
    'k_edc_pair_segsum': 4 * EDC_LEN + 4 * EDC_LEN,                            # direct path in, composed window samples out
    'k_edc_pair_seg_fwd': 3 * 4 * EDC_LEN,                                     # x, target EDC in, staged terms out
    'k_edc_pair_seg_bwd': 3 * 4 * EDC_LEN,                                     # staged terms, x in, gradient (window) out
    'k_lin_gamma_dots': 4 * EDC_LEN,                                           # EDC gradient (window) in (the G sums out: per band)
    # round 5: the EDC term in one launch per receiver (csrc/edcone.hip) and the light sum over the band's receivers
    'k_edc_lin_one': 3 * 4 * EDC_LEN,                                          # direct path, target EDC in; dL/dx (window) out
    'k_lin_gamma_win': 4 * EDC_LEN,                                            # dL/dx (window) in (the G sums out: per band)
    # (kernels of the stored-signal paths: FusedBankStep.spectral_edr / linear_transforms = False)
    'k_blu_col128_fwd': (8 * KU + 4 * 65536 + 4 * K + 4 * 65536) // 2,
    'k_blu_row512': 2 * 8 * 65536 // 2,
    'k_blu_col128_inv': (4 * 65536 + 4 * K + 4 * 65536 + 8 * KU) // 2,
    'k_stft4k_pair_power': 4 * K + 4 * FRAMES * NF,
    'k_stft4k_pair_power_bwd': (4 * K + 4 * FRAMES * NF // 2 + 4 * K + 4 * K),
    'k_edr_loss_cols': 3 * 4 * FRAMES * NF,
    'k_lin_combine_fwd': 4 * K + 4 * K,
}
ROOFLINE_EAGER_STEPS = 20
CPU_BASELINE_THREADS = 16            # the torch CPU path anti-scales beyond this on the 2x64-core host
PROFILE_TAG = 'r06'
REFERENCE_EPOCH_S_PER_BAND = 139.1   # reference trainer, N = 16, 8 vCPU, one band, one epoch


def host_cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def collective_version():
    try:
        v = torch.cuda.nccl.version()            # (on ROCm this is RCCL's version)
        return '.'.join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:                       # noqa: BLE001 -- informational only
        return f'unknown ({type(e).__name__})'


def library_stamp():
    """What the loaded C-ABI library was built from: the stamp __graft_entry__.build() wrote beside it (digest of the
    sources, compiler) checked against the sha256 of the file this process loaded and the sources in the tree."""
    import __graft_entry__ as ge
    from diffgfdn_amd import _lib
    stamp = ge.build_stamp() or {}
    lib_sha = ge._sha256(_lib.LIB_PATH)
    return {'hipcc': stamp.get('hipcc'), 'flags': stamp.get('flags'), 'lib_sha256': lib_sha,
            'stamp_matches_library': stamp.get('lib_sha256') == lib_sha,
            'stamp_matches_sources': stamp.get('sources_sha256') == ge.source_digest()}


def octave_band_response(centre_hz: float, fs: float, nfft: int, numtaps: int = 2049) -> np.ndarray:
    """(K,) response of a linear-phase octave band-pass FIR (stands in for the pyfar taps the
    reference loads, trainer.py:116-128 -- the taps are input data on this path)."""
    from scipy.signal import firwin
    lo, hi = centre_hz / np.sqrt(2.0), centre_hz * np.sqrt(2.0)
    taps = firwin(numtaps, [lo, hi], pass_zero=False, fs=fs)
    return np.fft.rfft(taps, n=nfft)


def trainer_config(centre_hz: float, max_epochs: int = 20, use_mask: bool = True, train_dir='/tmp/gfdn_bench/train'):
    from diffgfdn_amd.config import SubbandProcessingConfig, TrainerConfig
    return TrainerConfig(batch_size=BATCH, num_freq_bins=NFFT, max_epochs=max_epochs, lr=1e-3, io_lr=1e-2,
                         use_edc_mask=use_mask, use_colorless_loss=True, edc_loss_weight=10,
                         sparsity_loss_weight=2, use_asym_spectral_loss=True, device='cuda',
                         train_dir=train_dir, ir_dir='/tmp/gfdn_bench/ir',
                         subband_process_config=SubbandProcessingConfig(
                             centre_frequency=centre_hz, frequency_range=(63, 8000), num_fraction_octaves=1))


# ``--recipe reference``: the sub-band driver's OWN configuration (src/run_subband_training_treble.py:61-73, :105-154, :392):
# eight octave bands 63 Hz ... 8 kHz, N = 12 = 3 groups x 4 lines, every band its own gain network
REFERENCE_RECIPE_CENTRES = (63.0, 125.0, 250.0, 500.0, 1000.0, 2000.0, 4000.0, 8000.0)
RECIPE = None


def recipe_network(centre_hz: float):
    """(hidden layers, neurons per layer) of a band's gain network"""
    if RECIPE != 'reference':
        return 5, 16                     # (BASELINE.md's homogeneous recipe: the headline)
    f = int(round(centre_hz))
    if f == 63:
        return 1, 8
    if f == 125:
        return 1, 16
    if f in (250, 500, 1000):
        return 5, 16
    return 3, 128


def recipe_network_of(recipe: str, centre_hz: float):
    saved = RECIPE
    globals()['RECIPE'] = recipe
    try:
        return recipe_network(centre_hz)
    finally:
        globals()['RECIPE'] = saved


def build_net(device, delays, common_decay_times, centre_hz: float = 500.0):
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    hidden, neurons = recipe_network(centre_hz)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=hidden, num_neurons_per_layer=neurons,
                            num_fourier_features=20)
    return DiffGFDNVarReceiverPos(FS, G, delays, device, fl, of, use_absorption_filters=False,
                                  common_decay_times=common_decay_times, use_colorless_loss=True).to(device)


def build_workload(device, seed: int, num_receivers: int = NUM_RECEIVERS, centre_hz: float = 500.0,
                   room_seed: int = 0, make_trainer: bool = True, t60max: float = 1.5):
    from diffgfdn_amd.config import DiffGFDNConfig
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset, split_dataset
    from diffgfdn_amd.synthetic import synthetic_room
    from diffgfdn_amd.trainer import VarReceiverPosTrainer

    room = synthetic_room(num_receivers, G, FS, 64000, seed=room_seed, t60_range=(0.3, t60max))
    ds = RoomDataset(G, FS, room['source_position'], room['receiver_position'], room['rirs'],
                     room['common_decay_times'], nfft=NFFT, device=device)
    data = MultiRIRDataset(device, ds)
    torch.manual_seed(seed)
    np.random.seed(seed)
    cfg = DiffGFDNConfig(num_groups=G, num_delay_lines=G * NPER, sample_rate=FS, seed=23463 + int(centre_hz))
    delays = cfg.delay_length_samps
    net = build_net(device, delays, room['common_decay_times'], centre_hz)
    tc = trainer_config(centre_hz)
    filt = torch.tensor(octave_band_response(centre_hz, FS, NFFT), device=device).to(torch.complex64)
    train_idx, valid_idx, test_idx = split_dataset(data, 0.8, test_ratio=0.1)
    if not make_trainer:
        return room, data, net, tc, (train_idx, valid_idx), filt, delays
    pg = dist.group.WORLD if dist.is_initialized() else None
    trainer = VarReceiverPosTrainer(net, tc, subband_filter_freq_resp=filt, process_group=pg,
                                    capturable=True)
    start, length = trainer.criterion[1].window(K)
    data.precompute_decay_targets(WIN, start, length)
    return room, data, net, trainer, (train_idx, valid_idx), filt, delays


def band_t60max(nbands: int, distinct: bool):
    """Longest decay time of every band's synthetic room: 1.5 s for all, or -- ``--distinct-t60`` --
    falling from 1.5 s in the lowest band to 0.6 s in the highest, as measured octave bands do: every band then has its
    own EDC window (reference trainer.py:56-59)."""
    if not distinct or nbands == 1:
        return [1.5] * nbands
    return [float(v) for v in np.round(np.linspace(1.5, 0.6, nbands), 3)]


def build_bank_workload(device, seed: int, centres, num_receivers: int = NUM_RECEIVERS, max_epochs: int = 20,
                        train_dir='/tmp/gfdn_bench/train', t60max=None):
    """One model + dataset per octave band (run_subband_training_treble.py:175-204), stacked into a
    band bank.  Returns the 500 Hz band's room / delays / filter for the CPU baseline leg."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    nets, datasets, filts, splits = [], [], [], []
    cpu_leg = None
    for q, f in enumerate(centres):
        room, data, net, tc, split, filt, delays = build_workload(
            device, seed + q, num_receivers, centre_hz=f, room_seed=q, make_trainer=False,
            t60max=1.5 if t60max is None else t60max[q])
        nets.append(net)
        datasets.append(data)
        filts.append(filt)
        splits.append(split)
        if cpu_leg is None or f == 500.0:          # the CPU baseline leg times the 500 Hz band
            cpu_leg = (room, delays, filt)
    bank = BandBank(nets)
    pg = dist.group.WORLD if dist.is_initialized() else None
    tc = trainer_config(500.0, max_epochs, train_dir=train_dir)
    trainer = BandBankTrainer(bank, tc, subband_filter_freq_resp=torch.stack(filts), process_group=pg,
                              band_names=[int(f) for f in centres])
    sds = BandStackedDataset(datasets, free_sources=True)
    sds.precompute_decay_targets(WIN, *trainer._target_window(K))
    for d in datasets:                       # only the stacked stores are read from here on
        d.rir_mag_response = d.late_rir_mag_response = None
    sds.rir_mag_response = torch.empty((1, K), dtype=torch.complex64, device=device)   # shape carrier
    torch.cuda.empty_cache()
    return cpu_leg, sds, bank, trainer, splits


def hip_step_at(init, rirs, pos, delays, room, filt_np, keep, device, want_objects: bool = False):
    """Loss terms and gradients of the TIMED path (one-band bank, explicit step, slot order, pair-interleaved signals)
    at the parameters / batch / EDC mask the CPU baseline starts from: the other side of ``loss_delta_vs_cpu``."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset
    net = build_net(device, delays, room['common_decay_times'])
    with torch.no_grad():
        net.input_gains.copy_(init['input_gains'])
        net.output_gains.copy_(init['output_gains'])
        net.feedback_loop.M.copy_(init['M'])
    net.output_scalars.mlp.load_state_dict(init['mlp'])
    ds = MultiRIRDataset(device, RoomDataset(G, FS, room['source_position'], pos, rirs.copy(),
                                             room['common_decay_times'], nfft=NFFT, device=device))
    # the CPU leg normalises positions over the WHOLE grid (the dataset's own normalisation spans only this batch)
    ds.norm_listener_position = torch.tensor(init['norm_pos'], device=device)
    bank = BandBank([net])
    filt = torch.tensor(filt_np, device=device).to(torch.complex64).reshape(1, -1)
    tr = BandBankTrainer(bank, trainer_config(500.0, 1), subband_filter_freq_resp=filt, band_names=[500])
    sds = BandStackedDataset([ds])
    start, length = tr._decay_window(K)
    sds.precompute_decay_targets(WIN, start, length)
    B = rirs.shape[0]
    maskw = torch.zeros(length, dtype=torch.float32, device=device)
    maskw[keep.reshape(-1).to(device)] = 1.0 / (B * keep.numel())
    batch = sds.collate(sds.global_rows([list(range(B))]))
    out = tr._fused.run(batch, maskw, 1.0, normalize_first=True, train=True, opt_step=False)
    torch.cuda.synchronize()
    losses = {k: float(v) for k, v in out.items() if k.endswith('_loss')}
    opt = tr.optimizer
    grads, off = {}, 0
    names = {id(bank.output_gains): 'output_gains', id(bank.input_gains): 'input_gains',
             id(bank.output_scalars_w): 'mlp', id(bank.feedback_loop_M): 'M'}
    for p in opt._params:
        grads[names[id(p)]] = opt.flat_grad[off:off + p.numel()].detach().cpu().double().numpy().copy()
        off += p.numel()
    if want_objects:
        return losses, grads, tr, bank, sds
    return losses, grads


def cpu_reference_step(room, delays, filt_np, steps: int = 2):
    """``steps`` + 1 optimiser steps of ONE band on the host cores with the CPU oracle (the first is the warm-up and the one
    whose loss terms and gradients are kept).  Returns a dict: times (s per step), first (loss terms), keep0 (EDC mask),
    grads0 (gradients at the initial, normalized parameters), init (those parameters), rirs / pos (the batch), cores."""
    from oracle import gfdn_oracle as orc
    from oracle.cpu_trainer import OracleGridTrainer
    cores = min(CPU_BASELINE_THREADS, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    rng = np.random.RandomState(1)
    idx = rng.permutation(room['rirs'].shape[0])[:BATCH]
    rirs = room['rirs'][idx].copy()
    pos = room['receiver_position']
    npos = (pos - pos.min(0)) / ((pos.max(0) - pos.min(0)) + 1e-12)
    mix, win = int(20e-3 * FS), int(5e-3 * FS)
    w = np.hanning(win)
    full = np.fft.rfft(rirs, n=NFFT, axis=-1)
    early = rirs[:, :mix].copy()
    early[:, -(win // 2):] *= w[win // 2:]
    batch = {'z_values': torch.tensor(np.exp(1j * 2 * np.pi * np.fft.rfftfreq(NFFT))),
             'norm_listener_position': torch.tensor(npos[idx]),
             'listener_position': torch.tensor(pos[idx]),
             'target_early_response': torch.tensor(np.fft.rfft(early, n=NFFT, axis=-1)),
             'target_rir_response': torch.tensor(full)}
    torch.manual_seed(0)
    from diffgfdn_amd.dnn import MLP
    mlp = MLP(120, 5, 16, G, 1, 1)
    lins = [m for m in mlp.model if isinstance(m, torch.nn.Linear)]
    norms = [m for m in mlp.model if isinstance(m, torch.nn.LayerNorm)]
    lin = [(m.weight.detach().clone(), m.bias.detach().clone()) for m in lins]
    norm = [(m.weight.detach().clone(), m.bias.detach().clone()) for m in norms]
    N = G * NPER
    p = orc.GridModelParams(FS, delays, G, (2 * torch.randn(N, 1) - 1) / N, (2 * torch.randn(N, 1) - 1) / N,
                            (2 * torch.rand(G, NPER, NPER) - 1) / np.sqrt(NPER), torch.zeros(G * (G - 1) // 2),
                            room['common_decay_times'], lin, norm, 20)
    init = {'input_gains': p.input_gains.detach().clone(), 'output_gains': p.output_gains.detach().clone(),
            'M': p.M.detach().clone(), 'mlp': {k: v.detach().clone() for k, v in mlp.state_dict().items()},
            'norm_pos': npos[idx]}
    tr = OracleGridTrainer(p, lr=1e-3, io_lr=1e-2, edr_weight=1.0, edc_weight=10.0, spectral_weight=1.0,
                           sparsity_weight=2.0, use_asym=True, subband_filter=torch.tensor(filt_np))
    L = min(orc.ms_to_samps(float(np.max(room['common_decay_times'])) * 1e3, FS), K) - mix
    times, first, keep0, grads0 = [], None, None, None
    for s in range(steps + 1):           # first step is warm-up (allocator, thread pools)
        mask = torch.argwhere(torch.bernoulli(torch.empty(L).uniform_(0, 1)))
        t0 = time.time()
        tr.normalize(batch)
        _, terms = tr.train_step(batch, mask)
        times.append(time.time() - t0)
        if first is None:                # loss terms and gradients at the initial (normalized) parameters
            first, keep0 = terms, mask
            # packed like the bank's gain-network leaf: [W, b, gamma, beta] per hidden layer, then the output layer
            packed = []
            for i, (wl, bl) in enumerate(lin):
                packed += [wl.grad.reshape(-1), bl.grad.reshape(-1)]
                if i < len(norm):
                    packed += [norm[i][0].grad.reshape(-1), norm[i][1].grad.reshape(-1)]
            grads0 = {'input_gains': p.input_gains.grad.reshape(-1).double().numpy().copy(),
                      'output_gains': p.output_gains.grad.reshape(-1).double().numpy().copy(),
                      'M': p.M.grad.reshape(-1).double().numpy().copy(),
                      'mlp': torch.cat(packed).double().numpy().copy()}
    return {'times': times, 'first': first, 'keep0': keep0, 'grads0': grads0, 'init': init, 'rirs': rirs, 'pos': pos[idx],
            'cores': cores}


def cpu_baseline(room, delays, filt_np, steps: int = 2, device=None):
    """The same optimiser step on the host cores with the CPU oracle (reference restatement); with ``device`` also
    the loss terms and gradients of the timed HIP path on the same parameters, batch and EDC mask
    (``loss_delta_vs_cpu``, the metric's "EDR loss delta vs ref")."""
    ref = cpu_reference_step(room, delays, filt_np, steps)
    times, first, keep0, grads0, init, rirs, cores = (ref[k] for k in ('times', 'first', 'keep0', 'grads0', 'init', 'rirs',
                                                                         'cores'))
    pos_idx = ref['pos']
    sec = float(np.mean(times[1:]))
    out = {'value': BATCH * FRAMES / sec, 'unit': 'RIR-frames/s', 'cores': cores, 'kind': 'port',
           'host_cpu': host_cpu_model(), 'host_logical_cpus': os.cpu_count(),
           'sample': f'host CPU {host_cpu_model()} ({os.cpu_count()} logical CPUs, {cores} threads used); '
                     f'{steps} optimiser steps (normalize + fwd + EDR/EDC/colorless losses + bwd + Adam) of ONE band '
                     f'(500 Hz) of the same workload, batch {BATCH}, after 1 warm-up step; {sec:.2f} s/step.  The CPU '
                     'step recomputes the target EDR / EDC of its batch every step, as the reference does; the GPU '
                     'step reads them from stores precomputed once per dataset before the timed region '
                     '(SURVEY §8d "targets precomputed once")',
           'sec_per_step': sec}
    if device is not None:
        ours, g_hip = hip_step_at(init, rirs, pos_idx, delays, room, filt_np, keep0, device)
        delta = {k: {'cpu': first[k], 'hip': ours[k], 'rel': abs(ours[k] - first[k]) / abs(first[k])}
                 for k in ('edr_loss', 'edc_loss', 'spectral_loss', 'sparsity_loss')}
        tot_c, tot_h = sum(first.values()), sum(ours.values())
        delta['total'] = {'cpu': tot_c, 'hip': tot_h, 'rel': abs(tot_h - tot_c) / abs(tot_c)}
        gd = {}
        for k, gc in grads0.items():
            gh = g_hip[k]
            gd[k] = {'norm_cpu': float(np.linalg.norm(gc)), 'norm_hip': float(np.linalg.norm(gh)),
                     'max_dev_rel': float(np.abs(gh - gc).max() / (np.abs(gc).max() + 1e-300))}
        delta['gradients'] = gd
        delta['note'] = ('HIP = the timed path (explicit bank step, slot order, pair-interleaved signals) at the CPU '
                         "step's parameters, batch and EDC time mask (the masked EDC term is included)")
        out['loss_delta_vs_cpu'] = delta
    return out


# ---------------------------------------------------------------------------------------------------------------
# roofline helpers: committed profiles -> traffic / per-kernel table
# ---------------------------------------------------------------------------------------------------------------
def _profile(name):
    path = os.path.join(ROOT, 'profiles', f'{PROFILE_TAG}_{name}')
    return path if os.path.exists(path) else None


def pmc_traffic_bytes(kernel: str, table: str = 'pmc_hbm_bytes.csv'):
    """2 x FETCH_SIZE + WRITE_SIZE per launch of ``kernel`` from the committed rocprofv3 --pmc summary (gfx950
    correction as MI355X_MICROARCH.md prescribes), for the grid the step launches (the row with most launches)."""
    path = _profile(table)
    if path is None:
        return None
    rows = [r for r in csv.DictReader(open(path)) if r['kernel'].split('<')[0] == kernel]
    if not rows:
        return None
    r = max(rows, key=lambda r_: int(r_['launches']))
    return int(float(r['hbm_traffic_MB']) * 1e6)


def replayed_durations():
    """{kernel: average microseconds inside the replayed step} from the committed kernel trace, or None"""
    path = _profile('step_kernel_durations.csv')
    if path is None:
        return None
    return {r['kernel']: float(r['avg_us_in_step']) for r in csv.DictReader(open(path))}


def replayed_pair_window():
    """first start -> last end of the pair in ONE replayed step (profiles/<tag>_graph_step_timeline.txt), microseconds"""
    path = _profile('graph_step_timeline.txt')
    if path is None:
        return None
    lo, hi = None, None
    for line in open(path):
        f = line.split()
        if len(f) >= 6 and any(k in line for k in PAIR_KERNELS):
            try:
                a, b = float(f[0]), float(f[1])
            except ValueError:
                continue
            lo, hi = (a if lo is None else min(lo, a)), (b if hi is None else max(hi, b))
    return None if lo is None else hi - lo


def step_traffic():
    """PMC traffic (2 x FETCH_SIZE + WRITE_SIZE) summed over the launches of ONE replayed 7-band step, from the committed
    profiles (tools/make_profiles.py writes it beside the per-kernel table) or None"""
    path = _profile('step_traffic.json')
    if path is None:
        return None
    with open(path) as f:
        return json.load(f)


# kernels whose unit is a SIGNAL of the launch, not a RIR: in the linear step (the default) the transforms run on the bands'
# G group signals (nbands x G per launch), on the stored-signal path on the receivers' signals
SIGNAL_KERNELS = ('k_blu_col128_fwd', 'k_blu_row512', 'k_blu_col128_inv')


def roofline_top(units: int, n: int = 10, signals: int = None):
    """The kernels of the REPLAYED step by time per step (profiles/<tag>_step_kernel_durations.csv: durations inside 100
    consecutive graph replays, nothing else) with algorithmic bytes, measured traffic and fractions; the whole-run
    rocprofv3 average (which also covers host-launched, isolated and one-off launches) is kept beside each."""
    path, whole = _profile('step_kernel_durations.csv'), _profile('bench_kernel_stats.csv')
    if path is None:
        return None
    run_avg = {}
    if whole is not None:
        for r in csv.DictReader(open(whole)):
            run_avg.setdefault(r['Name'].split('(')[0].replace('void ', '').split('<')[0], float(r['AverageNs']) / 1e3)
    out = []
    for r in csv.DictReader(open(path)):
        name = r['kernel']
        if not name.startswith('k_'):
            continue
        us = float(r['avg_us_in_step'])
        e = {'kernel': name, 'launches_per_step': float(r['launches_per_step']), 'in_step_us': us,
             'us_per_step': float(r['us_per_step']), 'whole_run_avg_us': run_avg.get(name)}
        per = ALG_BYTES_PER_UNIT.get(name)
        if per is not None:
            alg = per * (signals if (signals is not None and name in SIGNAL_KERNELS) else units)
            e.update({'alg_bytes_per_launch': alg, 'achieved_GBs': alg / us / 1e3, 'frac': alg / us / 1e3 / HBM_PEAK_GBS})
            tr = pmc_traffic_bytes(name)
            if tr:
                e.update({'traffic': tr, 'traffic_over_alg': tr / alg})
        out.append(e)
        if len(out) == n:
            break
    return {'source': f'profiles/{PROFILE_TAG}_step_kernel_durations.csv (in-step durations) + profiles/{PROFILE_TAG}_pmc_hbm_bytes.csv'
                      f' + profiles/{PROFILE_TAG}_bench_kernel_stats.csv (whole-run averages)',
            'kernels': out}


def isolated_kernel_us(device, data, trainer, rows, nbands, iters: int = 30):
    """The dominant kernel alone on the chip, on the step's own stores (the transformed direct paths' STFT, the target EDR)
    and a batch of the step's shape; HIP events on its stream."""
    from diffgfdn_amd import hip_ops as ops
    tiled = trainer._fused.tiled_spectra
    Sd = data.direct_stft(trainer.subband_filter_freq_resp, K, WIN, tiled=tiled)
    idx = torch.as_tensor(rows, dtype=torch.long, device=device)
    items = idx.numel()
    Stau = torch.view_as_complex(torch.randn(nbands * G, FRAMES, NF, 2, device=device) * 0.01)
    rgain = torch.rand(items, G, device=device)
    nch = ops.lin_gamma_dots_tiles(K)
    one = trainer._fused.edr_one_launch
    parts = torch.empty((items * G, nch + ops.edr_lin_parts(NF, fused=one)),
                        dtype=torch.float32, device=device)
    T_edr, sum_abs = (data.edr_target_tiled() if tiled else data.edr_store[1]), data.edr_store[2]
    ops.kernel_timer.watch = DOMINANT_KERNEL
    ops.kernel_timer.start()
    for _ in range(iters):
        if one:
            ops.edr_lin_loss_gsum(Sd, idx, Stau, rgain, nbands, T_edr, sum_abs, 1.0, dots=parts, col0=nch, tiled=tiled,
                                  nsplit=trainer._fused._edr_runs(nbands, items // nbands))
        else:
            ops.edr_lin_loss(Sd, idx, Stau, rgain, nbands, T_edr, sum_abs, 1.0, True, dots=parts, col0=nch, tiled=tiled)
    return ops.kernel_timer.stop()


def isolated_edc_us(device, data, trainer, rows, nbands, iters: int = 30):
    """k_edc_lin_one alone on the chip on the step's own stores (transformed direct paths, target EDC)"""
    from diffgfdn_amd import hip_ops as ops
    xd = data.direct_time(trainer.subband_filter_freq_resp, K)
    idx = torch.as_tensor(rows, dtype=torch.long, device=device)
    items = idx.numel()
    decay = torch.exp(-torch.arange(K, device=device) / 9000.0)
    tau = torch.randn((nbands * G + 1) // 2, K, 2, device=device) * 0.01 * decay[None, :, None]
    rgain = torch.rand(items, G, device=device)
    start, length = trainer._decay_window(K)
    T_edc = data.edc_store[1]
    maskw = torch.full((length,), 1.0 / (items * length), dtype=torch.float32, device=device)
    parts = torch.empty((items * G, 10), dtype=torch.float32, device=device)
    item_len, _ = trainer._item_windows(K, items // nbands, device)
    if item_len is not None:
        maskw = maskw.repeat(nbands, 1).contiguous()
    ops.kernel_timer.watch = 'k_edc_lin_one'
    ops.kernel_timer.start()
    for _ in range(iters):
        ops.edc_lin_one(xd, idx, tau, rgain, nbands, K, start, length, T_edc, maskw, 1.0, 10.0, True, trows=idx,
                        item_len=item_len, dots=parts, col=0)
    return ops.kernel_timer.stop()


def run_directional(args, device, rank, world):
    """BASELINE.json configs[3]: directional DiffGFDN, 2nd-order ambisonics (9 SH channels per group), 7 bands, 838
    receivers -- G = 3 groups x 9 lines (N = 27), J = 12 directions, batch 32 receivers per band and step; every
    band's step (SH-domain forward, directional responses, directional EDC loss, backward, Adam) is one HIP-graph
    replay, the bands run one after another (reference: DirectionalFDNVarReceiverPosTrainer, trainer.py:690-921).
    The analysis matrix and the common-slope amplitudes are synthetic inputs (spaudiopy / the dataset are absent)."""
    from diffgfdn_amd import hip_ops
    from diffgfdn_amd.config import (CouplingMatrixType, DiffGFDNConfig, FeedbackLoopConfig, OutputFilterConfig,
                                     TrainerConfig)
    from diffgfdn_amd.model import DiffDirectionalFDNVarReceiverPos
    from diffgfdn_amd.trainer import DirectionalFDNVarReceiverPosTrainer
    Gd, order, J, R = 3, 2, 12, args.receivers
    L = (order + 1) ** 2
    nb = args.bands
    rng = np.random.RandomState(7)
    z = torch.exp(1j * np.pi * torch.arange(K, dtype=torch.float64) / (K - 1)).to(device)
    steps = []
    for q in range(nb):
        torch.manual_seed(500 + q)
        delays = DiffGFDNConfig(num_groups=Gd, num_delay_lines=Gd * L, sample_rate=FS, seed=23463 + q).delay_length_samps
        fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
        of = OutputFilterConfig(use_svfs=False, num_hidden_layers=5, num_neurons_per_layer=16, num_fourier_features=20)
        A = rng.randn(J, L).astype(np.float32)
        net = DiffDirectionalFDNVarReceiverPos(FS, Gd, delays, device, fl, of, ambi_order=order,
                                               common_decay_times=np.linspace(0.5, 1.4, Gd)[None, :],
                                               use_colorless_loss=True, analysis_matrix=A).to(device)
        tc = TrainerConfig(use_colorless_loss=True, use_asym_spectral_loss=True, edc_loss_weight=10.0,
                           sparsity_loss_weight=2.0, use_edc_mask=False, lr=1e-3, io_lr=1e-2, device='cuda',
                           train_dir='/tmp/gfdn_bench/dir_t', ir_dir='/tmp/gfdn_bench/dir_a')
        tr = DirectionalFDNVarReceiverPosTrainer(net, tc, capturable=True)
        store = {'listener_position': torch.tensor(rng.uniform(0, 10, (R, 3)), device=device),
                 'norm_listener_position': torch.tensor(rng.uniform(0, 1, (R, 3)), device=device),
                 'target_common_slope_amps': torch.tensor(rng.uniform(0.1, 1.0, (R, J, Gd)), device=device)}
        idx = torch.arange(BATCH, device=device)
        batch = {'z_values': z, 'source_position': torch.zeros(BATCH, 3, device=device, dtype=torch.float64)}
        batch.update({k: v[idx].clone() for k, v in store.items()})
        steps.append((tr, batch if args.dir_bank else tr.graphed(batch).capture(), store))
    gen = torch.Generator().manual_seed(100 + rank)
    bank = None
    if args.dir_bank:
        # trainer.DirectionalBank: ONE graph steps all bands (dealt onto ``--dir-streams`` lanes inside the capture)
        from diffgfdn_amd.trainer import DirectionalBank
        bank = DirectionalBank([t for t, _, _ in steps], [b for _, b, _ in steps], lanes=args.dir_streams).capture()
        steps = [(t, st, store) for (t, _, store), st in zip(steps, bank.steps)]

    # the bands are independent models: their graph replays go round-robin onto ``--dir-streams`` streams, so that one
    # band's latency-bound stretches (a dozen few-microsecond launches, the serial expm adjoint) run beside another's
    # bandwidth-bound transforms
    nstreams = max(1, min(args.dir_streams, nb))
    lanes = [torch.cuda.Stream(device=device) for _ in range(nstreams)] if nstreams > 1 else None

    def one_step():
        main = torch.cuda.current_stream()
        if bank is not None:
            for tr, step, store in steps:
                sel = torch.randperm(R, generator=gen)[:BATCH].to(device)
                for k, v in store.items():
                    step.batch[k].copy_(v.index_select(0, sel))
            return bank()[-1]
        for i, (tr, step, store) in enumerate(steps):
            sel_host = torch.randperm(R, generator=gen)[:BATCH]
            if lanes is None:
                sel = sel_host.to(device)
                for k, v in store.items():
                    step.batch[k].copy_(v.index_select(0, sel))
                out = step()
            else:
                lane = lanes[i % nstreams]
                lane.wait_stream(main)
                with torch.cuda.stream(lane):
                    # the indices are created ON the lane that reads them: allocated on the main stream their block would
                    # return to main's pool while the lane's index_select is still queued behind the previous replay, and
                    # a later upload could overwrite it (the bug class bandbank.py documents for cross-stream tensors)
                    sel = sel_host.to(device)
                    for k, v in store.items():
                        step.batch[k].copy_(v.index_select(0, sel))
                    out = step()
        if lanes is not None:
            for lane in lanes:
                main.wait_stream(lane)
        return out

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total, parts = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # roofline leg: the backward kernel of the directional EDC loss (the largest HBM-bound launch of the band-step since the
    # transforms run on the 27 line responses), in the step
    tr, step, store = steps[0]
    lines = tr._use_lines(step.batch)
    hip_ops.kernel_timer.watch = 'k_em_bwd' if lines else 'irfft_pow2_fwd'
    hip_ops.kernel_timer.start()
    for _ in range(10):
        tr.train_step(step.batch)
    kt = hip_ops.kernel_timer.stop()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rirs_per_s = nb * BATCH * world * args.steps / elapsed
    out = {'metric': 'RIR-frames/sec', 'value': rirs_per_s * FRAMES, 'unit': 'RIR-frames/s', 'n_gpus': world,
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': f'directional DiffGFDN: {nb} octave bands (independent models; ' + ('ONE graph for all bands, bands on ' if bank is not None else 'graph replays round-robin on ') + f'{nstreams} stream(s)), each {Gd} groups x {L} SH '
                                  f'channels (N = {Gd * L}), {J} directions, {R} receivers, nfft 131072, batch {BATCH} '
                                  'receivers/band/step/GPU; step = line responses + 27 line transforms + directional EDC + colorless losses + bwd + '
                                  'Adam per band', 'bank': bank is not None, 'bands': nb, 'receivers': R, 'band_streams': nstreams,
                       'directions': J, 'delay_lines': Gd * L, 'rirs_per_s': rirs_per_s,
                       'ms_per_band_step': 1e3 * elapsed / args.steps / nb,
                       'final_loss': float(total)}}
    if kt and lines:
        crit = tr.criterion[0]
        win = min(crit.edc_len_samps, NFFT - crit.mixing_time_samps)
        units = kt['units_per_launch']                      # receivers per launch
        per = 2 * L * win * 4                               # the receiver's 9 SH signals on the EDC window in, their gradient out
        us = kt['avg_ms'] * 1e3
        out['roofline'] = {'bound': 'hbm', 'achieved': units * per / us / 1e3, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': units * per / us / 1e3 / HBM_PEAK_GBS,
                           'traffic': pmc_traffic_bytes('k_em_bwd', 'directional_pmc_hbm_bytes.csv'),
                           'traffic_source': f'profiles/{PROFILE_TAG}_directional_pmc_hbm_bytes.csv (2 x FETCH_SIZE + WRITE_SIZE, '
                                             'separate --pmc runs)',
                           'kernel': 'k_em_bwd<9, 12> (directional EDC loss, backward: dL/dEDC recomputed, prefix scans of the 12 '
                                     'directions in registers, gradient mixed back to the 9 SH channels)',
                           'avg_launch_us': us, 'launches': kt['launches'], 'alg_bytes_per_unit': per,
                           'units_per_launch': units, 'edc_window_samples': win,
                           'measured': 'HIP events around the one kernel (gfdn_edc_loss_model_mixed_stages) in 10 host-launched '
                                       'steps'}
    elif kt:
        units = kt['units_per_launch']                      # SH-domain responses per launch (32 receivers x 9 channels)
        per = 8 * K + 2 * 8 * 65536 + 4 * NFFT              # spectrum in, work block out and in, samples out
        us = kt['avg_ms'] * 1e3
        ta, tb = (pmc_traffic_bytes(k, 'directional_pmc_hbm_bytes.csv') for k in ('k_pw_inv_a', 'k_pw_inv_b'))
        out['roofline'] = {'bound': 'hbm', 'achieved': units * per / us / 1e3, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': units * per / us / 1e3 / HBM_PEAK_GBS,
                           'traffic': (ta + tb) if (ta and tb) else None,
                           'traffic_source': f'profiles/{PROFILE_TAG}_directional_pmc_hbm_bytes.csv (2 x FETCH_SIZE + WRITE_SIZE of '
                                             'both passes, separate --pmc runs)',
                           'kernel': 'k_pw_inv_a + k_pw_inv_b (irfft, n = 131072, of the SH-domain responses: the two '
                                     'register-resident 256 x 256 passes)',
                           'avg_launch_us': us, 'launches': kt['launches'], 'alg_bytes_per_unit': per,
                           'units_per_launch': units, 'measured': 'HIP events around both passes in 10 host-launched steps'}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline_directional(steps[0][0], z, Gd, L, J)
    return out


def cpu_baseline_directional(tr, z, Gd, L, J, receivers: int = 2):
    """One band-step of the directional model on the host cores with the CPU oracle (oracle/cpu_trainer.py
    directional_band_step: forward, directional EDC loss, backward under autograd in float64 / complex128) on a bounded
    sample -- ``receivers`` of the 32 receivers of a band-step, the band's current parameters, timed after one warm-up
    call -- and the same sample's loss AND parameter gradients on the HIP path beside it.  The per-bin systems of the
    feedback loop are shared by a batch's receivers, so the rate of the 2-receiver sample is a LOWER bound of the CPU's rate
    on a full batch: it is reported as such (``value_is_lower_bound``), not as a like-for-like figure."""
    from oracle.cpu_trainer import directional_band_step
    cores = min(CPU_BASELINE_THREADS, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    net = tr.net
    rng = np.random.RandomState(3)
    pos = torch.tensor(rng.uniform(0, 1, (receivers, 3)))
    amps = torch.tensor(rng.uniform(0.1, 1.0, (receivers, J, Gd)))
    crit = tr.criterion[0]
    state = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    delays = [int(d) for d in net.delays.detach().cpu().reshape(-1).tolist()]
    nff = net.sh_output_scalars.num_fourier_features
    A = net.sh_output_scalars.analysis_matrix
    args_c = (state, delays, A.cpu().numpy(), z.cpu(), pos, amps, crit.envelopes.cpu(), Gd, L, nff, crit.mixing_time_samps,
              crit.edc_len_samps)
    directional_band_step(*args_c, edc_weight=tr.config.edc_loss_weight)          # warm-up (allocator, thread pools)
    t0 = time.time()
    loss_c, _, _, grads_c = directional_band_step(*args_c, edc_weight=tr.config.edc_loss_weight)
    sec = time.time() - t0
    torch.set_num_threads(min(4, os.cpu_count() or 1))
    dev = z.device
    batch = {'z_values': z, 'source_position': torch.zeros(receivers, 3, device=dev, dtype=torch.float64),
             'listener_position': (10 * pos).to(dev), 'norm_listener_position': pos.to(dev),
             'target_common_slope_amps': amps.to(dev)}
    net.zero_grad(set_to_none=True)
    if tr._use_lines(batch):              # (the timed route: the output stage behind the transform, csrc/dirlin.hip)
        Y, c, w = net.output_stage_inputs(batch)
        edc = crit.forward_lines(Y, c, w, Gd, L, None, A, batch['target_common_slope_amps'],
                                 weight=tr.config.edc_loss_weight)
    else:
        out = net(batch)
        H_sh = out[0] if isinstance(out, tuple) else out
        edc = crit.forward_sh(H_sh, A, batch['target_common_slope_amps'], weight=tr.config.edc_loss_weight)
    edc.backward()
    loss_h = float(edc.detach())
    gd = {}
    for name, p_ in net.named_parameters():
        if name in grads_c and p_.grad is not None:
            gc, gh = grads_c[name].double().numpy(), p_.grad.detach().cpu().double().numpy()
            gd[name] = {'norm_cpu': float(np.linalg.norm(gc)), 'norm_hip': float(np.linalg.norm(gh)),
                        'max_dev_rel': float(np.abs(gh - gc).max() / (np.abs(gc).max() + 1e-300))}
    net.zero_grad(set_to_none=True)
    return {'value': receivers * FRAMES / sec, 'unit': 'RIR-frames/s', 'cores': cores, 'kind': 'port',
            'value_is_lower_bound': True,
            'host_cpu': host_cpu_model(), 'host_logical_cpus': os.cpu_count(), 'sec_per_step': sec,
            'sample': f'host CPU {host_cpu_model()} ({os.cpu_count()} logical CPUs, {cores} threads used); ONE band-step '
                      f'(SH-domain forward, directional EDC loss, backward; no optimiser update) at {receivers} of the 32 '
                      f'receivers of a band-step, after one warm-up call; {sec:.1f} s.  The feedback-loop solve (65 537 bins x '
                      '27 lines) does not depend on the number of receivers, so this per-receiver rate is a LOWER bound of '
                      'the rate of a full 32-receiver batch -- not to be divided into the GPU figure',
            'loss_delta_vs_cpu': {'edc_loss': {'cpu': float(loss_c), 'hip': loss_h,
                                               'rel': abs(loss_h - float(loss_c)) / abs(float(loss_c))},
                                  'gradients': gd}}


def self_launch(args) -> int:
    """N > 1 without a launcher: start N fresh ranks (nothing in this process has touched the GPU)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--repeats', type=int, default=5,
                    help='timed regions of --steps steps each (N = 1; ms_per_step is the MEDIAN region, the spread is in '
                         'extra.repeat_spread): a 20-step region of this workload is an 8 ms sample')
    ap.add_argument('--scaling', choices=('auto', 'weak', 'strong'), default='auto',
                    help='auto (default): N = 1 the step at batch 32; N > 1 the headline is STRONG scaling (the reference\'s '
                         'global batch of 32 per band split over the ranks) with the weak figure (32 receivers per band per '
                         'RANK) and the band-sharded placement in "extra"; weak / strong: that one measurement only')
    ap.add_argument('--config', choices=('omni', 'directional'), default='omni',
                    help='omni: BASELINE.json configs[1]/[2] (the headline); directional: configs[3] (2nd-order ambisonics)')
    ap.add_argument('--epoch', action='store_true',
                    help='time whole epochs (19 train + 5 validation steps + checkpoints); --steps = epochs timed')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the N = 32 and directional sub-records of the default N = 1 line')
    ap.add_argument('--extra-steps', type=int, default=40, help='timed steps of each sub-record')
    ap.add_argument('--dir-bank', type=int, default=1,
                    help='directional config: 1 (default) = trainer.DirectionalBank, ONE graph steps all bands; 0 = one graph '
                         'per band, replayed round-robin on --dir-streams streams')
    ap.add_argument('--dir-streams', type=int, default=4,
                    help='--config directional: lanes the bands are dealt onto inside the bank\'s graph (--dir-bank 1), or '
                         'streams the bands\' own graph replays are spread over (--dir-bank 0); measured same-box: one graph '
                         'per band on 2 streams 0.56-0.58 ms per band-step, the bank on 2 / 3 / 4 / 7 lanes 0.48 / 0.45 / 0.44 / '
                         '0.44')
    ap.add_argument('--captured-allreduce', action='store_true',
                    help='N > 1: capture the RCCL all-reduce inside the step graph (when every rank can)')
    ap.add_argument('--eager', action='store_true', help='launch every kernel from the host (no HIP graph)')
    ap.add_argument('--pipe-steps', type=int, default=None,
                    help='steps per graph of the pipelined chain (even; 0: one step per graph); default: the trainer\'s')
    ap.add_argument('--per-step-copy', action='store_true',
                    help='hand every step its receivers by a host copy in front of the replay (diagnostic; default: the '
                         'batches go to the device as one schedule)')
    ap.add_argument('--cpu-steps', type=int, default=3)
    ap.add_argument('--receivers', type=int, default=NUM_RECEIVERS)
    ap.add_argument('--classic', action='store_true',
                    help='with --bands 1: the per-band VarReceiverPosTrainer path instead of a one-band bank')
    ap.add_argument('--lines-per-group', type=int, default=NPER,
                    help='delay lines per group (4 groups): 8 gives the N = 32 configuration')
    ap.add_argument('--distinct-t60', action='store_true',
                    help='every band gets its own longest decay time (1.5 s ... 0.6 s) and therefore its own EDC window, as '
                         'the reference\'s per-band datasets do (trainer.py:56-59)')
    ap.add_argument('--batch', type=int, default=BATCH,
                    help='receivers per band and step on one GPU (the reference trains with 32, trainer.py:373-379; other '
                         'values measure what a rank of a strong-scaling job steps: DESIGN.md section 7)')
    ap.add_argument('--recipe', choices=('baseline', 'reference'), default='baseline',
                    help='baseline: BASELINE.md\'s configuration (7 bands, N = 16, the 5 x 16 gain network in every band: the '
                         'headline); reference: the sub-band driver\'s own (run_subband_training_treble.py:61-73, :105-154, '
                         ':392: 8 bands, N = 12 = 3 x 4, per-band gain networks 1 x 8 / 1 x 16 / 5 x 16 / 3 x 128)')
    ap.add_argument('--bands', type=int, default=None,
                    help='octave bands stepped together (1 = BASELINE.json configs[1], the 500 Hz band alone)')
    args = ap.parse_args()
    if args.recipe == 'reference':
        globals().update(RECIPE='reference', G=3, BAND_CENTRES=REFERENCE_RECIPE_CENTRES)
    if args.bands is None:
        args.bands = len(BAND_CENTRES)
    if args.lines_per_group != NPER:
        globals()['NPER'] = args.lines_per_group
    if args.batch != BATCH:
        if args.batch < 2 or args.batch % 2:
            raise SystemExit("--batch: an even number of receivers per band (receiver pairs share a transform)")
        globals()['BATCH'] = args.batch
    if not 1 <= args.bands <= len(BAND_CENTRES):
        raise SystemExit(f"--bands must be 1..{len(BAND_CENTRES)}")
    if args.classic and args.bands != 1:
        raise SystemExit("--classic steps one band: use --bands 1")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # ---- launch contract, checked BEFORE any GPU call
    if 'WORLD_SIZE' not in os.environ:
        if args.gpus > 1:
            sys.exit(self_launch(args))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ['WORLD_SIZE'])
        rank = int(os.environ.get('RANK', '0'))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if args.gpus != world:
            raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    # host-side torch ops on tiny CPU tensors (index lists) crawl when the intra-op pool spans all 256 host cores
    torch.set_num_threads(min(4, os.cpu_count() or 1))
    # Rehearsal knobs for a ONE-GPU box (the driver's multi-GPU run uses neither): GFDN_BENCH_ONE_DEVICE=1 puts every
    # rank on cuda:0 and GFDN_BENCH_BACKEND=gloo carries the collectives (RCCL refuses two ranks on one device) --
    # the N > 1 code path (shared mask seed, loss slots in the bucket, two-graph step, max-over-ranks timing) end to end.
    dev_index = 0 if os.environ.get('GFDN_BENCH_ONE_DEVICE') else local_rank
    backend = os.environ.get('GFDN_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    ranks_seen, rank_devices = [0], [dev_index]
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
        seen = [None] * world
        dist.all_gather_object(seen, (rank, dev_index, socket.gethostname()))
        ranks_seen = sorted(r for r, _, _ in seen)
        assert ranks_seen == list(range(world)), ranks_seen
        rank_devices = [d for _, d, _ in sorted(seen)]

    if args.config == 'directional':
        out = run_directional(args, device, rank, world)
        if rank == 0:
            out['ranks_seen'] = ranks_seen
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    out = run_omni(args, device, rank, world, ranks_seen, rank_devices)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                 # rank 0 may still be in its roofline / baseline legs
        dist.destroy_process_group()


def timed_steps(step, draws, args, world, device, eager=False, per_step_copy=False, repeats=1, spread=None):
    """warm-up, then EXACTLY ``args.steps`` steps between barrier + synchronize on both sides; returns (seconds, max over
    the ranks; the last step's loss parts).  ``repeats`` > 1: that many such regions back to back (the same receivers), the
    MEDIAN region's seconds returned and min / max / median per step left in ``spread``."""
    warm, timed = draws
    scheduled = not eager and not per_step_copy and hasattr(step, 'run_schedule')

    def one_step(sel):
        if eager:                            # the same launch sequence as the graph, launched from the host
            step._load_inputs(sel)
            return step._eager()
        return step(sel)

    if scheduled:
        for parts in step.run_schedule(warm):
            pass
        # (the timed steps' receivers as ONE pinned int64 table, drawn up front like the lists were: the upload itself stays
        # inside the timed region, the conversion of 20 x 224 Python integers -- 0.2 ms of host time per region -- does not)
        timed = torch.tensor([list(sel) for sel in timed], dtype=torch.long).pin_memory()
    else:
        for sel in warm:
            one_step(sel)
    regions = []
    for _ in range(max(1, int(repeats))):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if scheduled:
            for parts in step.run_schedule(timed):           # (uploads the schedule inside the timed region)
                pass
        else:
            for sel in timed:
                parts = one_step(sel)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        regions.append(elapsed)
    if spread is not None:
        spread.update({'regions': len(regions), 'steps_per_region': len(timed),
                       'ms_per_step_min': 1e3 * min(regions) / len(timed), 'ms_per_step_max': 1e3 * max(regions) / len(timed),
                       'ms_per_step_median': 1e3 * float(np.median(regions)) / len(timed)})
    return float(np.median(regions)), parts


def run_band_sharded(args, device, rank, world, centres):
    """The reference's own parallel axis at its own batch size: whole bands over the ranks (rank r trains bands r, r + N,
    ... with the reference's batch of 32 receivers each, run_subband_training_treble.py:175-204 spread over GPUs) -- no
    collective at all.  Whole-job rate = bands x 32 receivers x steps / the slowest rank's time."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    mine = [(q, f) for q, f in enumerate(centres) if q % world == rank]
    step = draws = None
    vrirs, vsec = 0, 0.0
    if mine:
        nets, datasets, filts, splits = [], [], [], []
        tmax = band_t60max(len(centres), args.distinct_t60)
        for q, f in mine:
            room, data, net, tc, split, filt, delays = build_workload(device, 1234 + q, args.receivers, centre_hz=f,
                                                                      room_seed=q, make_trainer=False, t60max=tmax[q])
            nets.append(net), datasets.append(data), filts.append(filt), splits.append(split)
        bank = BandBank(nets)
        tr = BandBankTrainer(bank, trainer_config(500.0, 20, train_dir='/tmp/gfdn_bench/train_bs'),
                             subband_filter_freq_resp=torch.stack(filts), band_names=[int(f) for _, f in mine],
                             data_parallel=False)
        sds = BandStackedDataset(datasets, free_sources=True)
        sds.precompute_decay_targets(WIN, *tr._target_window(K))
        gen = torch.Generator().manual_seed(300 + rank)
        st = [torch.tensor(s_[0]) for s_ in splits]
        draw = lambda: sds.global_rows([t[torch.randperm(len(t), generator=gen)[:BATCH]].tolist() for t in st])
        step = tr.graphed(sds, BATCH)
        draws = ([draw() for _ in range(args.warmup)], [draw() for _ in range(args.steps)])
    if step is not None:
        elapsed, _ = timed_steps(step, draws, args, world, device)
    else:
        # more ranks than bands (7 bands on 8 GPUs): the HYBRID placement -- a rank without a band takes the purely parallel
        # side work of the reference's loop, validation passes (trainer.py:389-397: forward + losses, no gradient, no
        # exchange), on a band of its own copy while the others train; it keeps the training ranks' barriers
        vrirs = 0
        vtr = vbatch = None
        try:
            q = rank % len(centres)
            tmax = band_t60max(len(centres), args.distinct_t60)
            room, data, net, tc, split, filt, delays = build_workload(device, 1234 + q, args.receivers, centre_hz=centres[q],
                                                                      room_seed=q, make_trainer=False, t60max=tmax[q])
            vbank = BandBank([net])
            vtr = BandBankTrainer(vbank, trainer_config(500.0, 20, train_dir='/tmp/gfdn_bench/train_bs'),
                                  subband_filter_freq_resp=torch.stack([filt]), band_names=[int(centres[q])],
                                  data_parallel=False)
            vsds = BandStackedDataset([data], free_sources=True)
            vsds.precompute_decay_targets(WIN, *vtr._target_window(K))
            vbatch = vsds.collate(vsds.global_rows([list(split[1][:BATCH])]))
            vtr.valid_step(vbatch)                # (warm-up: lazy stores, kernels)
        except Exception as e:                    # noqa: BLE001 -- the side work must never cost the training ranks their line
            print(f'[bench] rank {rank}: validation side work not started ({type(e).__name__}: {e})', file=sys.stderr)
            vtr = None
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        # The side work must not sit on the training ranks' clock: this rank reports to the closing barrier AT ONCE
        # (asynchronously) and runs validation batches only until that barrier completes -- i.e. until every training rank
        # has finished its timed steps -- and its own interval does not enter the maximum over the ranks.  (Round 5 ran
        # ``steps`` host-launched validation steps between the two barriers: the line's time was max(training, validation).)
        closing = dist.barrier(async_op=True)
        if vtr is not None:
            while not closing.is_completed() and vrirs < 1000 * BATCH * args.steps:
                vtr.valid_step(vbatch)
                vrirs += BATCH
            torch.cuda.synchronize()
        vsec = time.perf_counter() - t0
        closing.wait()
        t = torch.zeros(1, device=device, dtype=torch.float64)         # (this rank trained nothing: no share in the maximum)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rirs = len(centres) * BATCH * args.steps / elapsed
    out = {'placement': f'whole bands over the ranks (rank r: bands r, r + {world}, ...), batch {BATCH} receivers per band, '
                        'no collective', 'ms_per_step': 1e3 * elapsed / args.steps, 'rirs_per_s': rirs,
           'value': rirs * FRAMES, 'unit': 'RIR-frames/s', 'bands_on_rank0': len(mine) if rank == 0 else None}
    if world > len(centres):
        # validation RIRs the band-less ranks processed inside the same interval (summed over them)
        v = torch.tensor([float(vrirs) if not mine else 0.0, float(vsec) if not mine else 0.0], device=device, dtype=torch.float64)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        out['hybrid'] = {'placement': f'{len(centres)} ranks train one band each, {world - len(centres)} rank(s) run validation '
                                      'passes (forward + losses of 32-receiver batches, host-launched) meanwhile: no exchange',
                         'validation_rirs_in_interval': float(v[0].item()),
                         'validation_rirs_per_s': (float(v[0].item()) / float(v[1].item())) if float(v[1].item()) > 0 else 0.0,
                         'training_rirs_per_s': rirs}
    return out


def run_omni(args, device, rank, world, ranks_seen, rank_devices, sub_record=False):
    """The headline workload (BASELINE.json configs[1] / [2], or [4] with --lines-per-group 8).  N = 1: one measurement.
    N > 1: the headline is STRONG scaling at the reference's global batch of 32 receivers per band (trainer.py:373-379),
    with the weak-scaling figure (32 receivers per band per RANK) and the band-sharded placement beside it in ``extra``."""
    from diffgfdn_amd import hip_ops
    nbands = args.bands
    use_bank = not args.classic
    scaling = args.scaling if args.scaling != 'auto' else ('strong' if world > 1 else 'weak')
    if scaling == 'strong' and BATCH % world:
        raise SystemExit(f"strong scaling splits the batch of {BATCH} receivers per band: {world} ranks do not divide it")
    b_local = BATCH if scaling == 'weak' else BATCH // world
    if not use_bank:
        room, data, net, trainer, split, filt, delays = build_workload(device, seed=1234,
                                                                      num_receivers=args.receivers)
        splits = [split]
        centres = (500.0,)
    else:
        centres = BAND_CENTRES[:nbands] if nbands > 1 else (500.0,)
        (room, delays, filt), data, net, trainer, splits = build_bank_workload(
            device, 1234, centres, args.receivers, max_epochs=args.warmup + args.steps if args.epoch else 20,
            t60max=band_t60max(len(centres), args.distinct_t60))

    epoch_info = None
    extra = {}
    step = None
    if args.epoch:
        # ---- whole epochs of the reference's loop (trainer.py:345-424) for all bands in lockstep
        if not use_bank:
            raise SystemExit("--epoch times the band bank")
        trainer.max_epochs = args.warmup + args.steps
        trainer.patience = 10 ** 9                               # (no early stop inside the measurement)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        trainer.train(data, [s[0] for s in splits], [s[1] for s in splits], batch_size=BATCH, log=False)
        torch.cuda.synchronize()
        ep = trainer.epoch_times[args.warmup:]
        t = torch.tensor([float(np.sum(ep))], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ntrain, nvalid = len(splits[0][0]), len(splits[0][1])
        tsteps = sum(1 for i0 in range(0, ntrain, BATCH) if min(BATCH, ntrain - i0) // world > 0)
        vsteps = sum(1 for i0 in range(0, nvalid, BATCH) if min(BATCH, nvalid - i0) // world > 0)
        rirs_epoch = nbands * sum((min(BATCH, ntrain - i0) // world) * world for i0 in range(0, ntrain, BATCH))
        steps_timed = len(ep) * tsteps
        ms_per_step = 1e3 * elapsed / max(steps_timed, 1)
        rirs_per_s = rirs_epoch * len(ep) / elapsed
        epoch_info = {'s_per_epoch_all_bands': elapsed / len(ep), 'epochs_timed': len(ep), 'train_steps': tsteps,
                      'valid_steps': vsteps, 'train_receivers': ntrain, 'valid_receivers': nvalid,
                      'checkpoints_per_epoch': nbands,
                      'reference_s_per_epoch_per_band': REFERENCE_EPOCH_S_PER_BAND,
                      'reference_s_per_epoch_all_bands': REFERENCE_EPOCH_S_PER_BAND * nbands,
                      'reference_note': 'BASELINE.md §2a: reference VarReceiverPosTrainer.train on 8 vCPU, N = 16, one '
                                        'band after another (wav export skipped)',
                      'speedup_vs_reference_epoch': REFERENCE_EPOCH_S_PER_BAND * nbands / (elapsed / len(ep))}
        total, ktimes, iso = torch.tensor(trainer.train_loss)[:, -1], {}, {}
        value = rirs_per_s * FRAMES
        scaling = 'strong'               # (an epoch is a fixed amount of work whatever the number of ranks)
    else:
        # every rank draws its own receivers (different shards of the global batch of every band)
        gen = torch.Generator().manual_seed(100 + rank)
        splits_t = [torch.tensor(s[0]) for s in splits]

        def draw_n(nrec):
            sel = [t[torch.randperm(len(t), generator=gen)[:nrec]].tolist() for t in splits_t]
            return sel[0] if not use_bank else data.global_rows(sel)

        # N > 1: the step is two graphs with the (eager) RCCL all-reduce of the gradient bucket between them unless
        # --captured-allreduce asks for the collective INSIDE one graph; that structure is taken only when every rank's
        # capture probe agrees (GraphedTrainStep.capture: MIN over the group), so the ranks can never be on different
        # structures.  The two-graph step is the default because it needs nothing of RCCL but an ordinary all-reduce.
        if world > 1:
            trainer.allreduce_in_graph = bool(args.captured_allreduce)
        step = trainer.graphed(data, b_local)      # normalize + train_step as ONE HIP-graph replay
        if args.pipe_steps is not None:
            step.pipe_steps = args.pipe_steps
        draws = ([draw_n(b_local) for _ in range(args.warmup)], [draw_n(b_local) for _ in range(args.steps)])
        spread = {}
        elapsed, parts = timed_steps(step, draws, args, world, device, eager=args.eager, per_step_copy=args.per_step_copy,
                                     repeats=(args.repeats if (world == 1 and not sub_record) else 1), spread=spread)
        if spread.get('regions', 1) > 1:
            extra['repeat_spread'] = spread
        total = parts['_total']

        # roofline leg (rank 0): the SAME launch sequence on the same streams, launched from the host so that the
        # dominant kernel can be bracketed by HIP events on its stream -- its duration in the step, beside its
        # duration alone on the chip
        ktimes, iso, kpair, iso_pair = {}, {}, {}, {}
        if rank == 0 and use_bank and world == 1 and getattr(trainer, '_fused', None) is not None and not sub_record:
            hip_ops.kernel_timer.watch = PAIR_KERNELS
            hip_ops.kernel_timer.start()
            for _ in range(ROOFLINE_EAGER_STEPS):
                batch = data.collate(draw_n(b_local))
                trainer._fused.run(batch, step.maskw, 1.0, normalize_first=True, train=True, opt_step=False)
            kpair = hip_ops.kernel_timer.stop_multi()
            iso_pair = {'k_edr_lin_wave': isolated_kernel_us(device, data, trainer, draw_n(b_local), nbands)}
            try:
                iso_pair['k_edc_lin_one'] = isolated_edc_us(device, data, trainer, draw_n(b_local), nbands)
            except Exception as e:                               # noqa: BLE001 -- a diagnostic leg
                print(f'[bench] isolated EDC launch not measured ({type(e).__name__}: {e})', file=sys.stderr)
            # the kernel the line names: the one that is LONGER in the REPLAYED step (committed kernel trace of the replayed
            # graph, profiles/<tag>_step_kernel_durations.csv; the host-launched steps of this leg order the two launches
            # differently); without a profile, the longer of this leg's brackets
            prof = replayed_durations()
            if prof and any(k in prof for k in PAIR_KERNELS):
                named = max((k for k in PAIR_KERNELS if k in prof and k in kpair), key=lambda k: prof[k], default=None)
            else:
                named = max((k for k in PAIR_KERNELS if k in kpair), key=lambda k: kpair[k]['avg_ms'], default=None)
            if named is not None:
                ktimes = dict(kpair[named])
                ktimes['event_pair_overhead_ms'] = iso_pair['k_edr_lin_wave'].get('event_pair_overhead_ms', 0.0)
                iso = iso_pair.get(named, {})
        ms_per_step = 1e3 * elapsed / args.steps
        rirs_per_s = nbands * b_local * world * args.steps / elapsed
        value = rirs_per_s * FRAMES

        if world > 1 and use_bank and args.scaling == 'auto':
            # ---- beside the strong-scaling headline: weak scaling (32 receivers per band per RANK, global batch 32 N) on
            # the same models and data, and the band-sharded placement (no collective)
            wstep = trainer.graphed(data, BATCH)
            wdraws = ([draw_n(BATCH) for _ in range(args.warmup)], [draw_n(BATCH) for _ in range(args.steps)])
            wel, _ = timed_steps(wstep, wdraws, args, world, device)
            wr = nbands * BATCH * world * args.steps / wel
            extra['weak'] = {'scaling': 'weak', 'batch_per_band_per_gpu': BATCH, 'global_batch_per_band': BATCH * world,
                             'ms_per_step': 1e3 * wel / args.steps, 'rirs_per_s': wr, 'value': wr * FRAMES,
                             'unit': 'RIR-frames/s'}
            del wstep
            extra['band_sharded'] = run_band_sharded(args, device, rank, world, centres)

    out = None
    if rank == 0:
        graph_mode = 'eager' if args.eager else 'hip_graph'
        if world > 1 and not args.eager and not args.epoch:
            graph_mode += ((f' (all-reduce captured: {step.collective_probe_nodes} graph node(s) in the capture probe)')
                           if step.allreduce_in_graph else ' (two graphs, eager all-reduce)')
        fused = getattr(trainer, '_fused', None)
        out = {
            'metric': 'RIR-frames/sec', 'value': value, 'unit': 'RIR-frames/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic', 'ranks_seen': ranks_seen,
            'config': {'workload': f'{nbands} octave-band GFDN(s) ({", ".join(str(int(f)) for f in centres)} Hz), each '
                                   f'N={G * NPER} ({G} groups x {NPER}) on a {args.receivers}-receiver grid, nfft 131072 '
                                   f'(K=65537), batch {b_local} receivers/band/step/GPU; step = normalize + fwd + '
                                   'EDR/EDC(mask)/colorless losses + bwd + [all-reduce] + Adam for every band'
                                   + (' (band bank: one launch per stage for all bands)' if use_bank else '')
                                   + ('; whole epochs: train + validation + checkpoints' if args.epoch else ''),
                       'bands': nbands, 'receivers': args.receivers, 'batch_per_band_per_gpu': b_local,
                       'rirs_per_step_per_gpu': nbands * b_local, 'global_batch_per_band': b_local * world,
                       'delay_lines': G * NPER, 'bins': K, 'rirs_per_s': rirs_per_s,
                       'band_t60max_s': band_t60max(len(centres), args.distinct_t60) if use_bank else [1.5],
                       'band_edc_windows': ((trainer._band_windows(K) or [trainer._decay_window(K)[1]] * nbands)
                                            if use_bank else [trainer._decay_window(K)[1]]),
                       'output_stage': (None if fused is None else
                                        ('time domain: x[b] = xd[row_b] + sum_g gain[b][g] irfft(T_g filt) -- G transforms '
                                         'per band and step instead of one per receiver' if fused.linear_transforms else
                                         'per receiver in the frequency domain, one transform per receiver pair')),
                       'edr_loss': (None if fused is None else
                                    ('on composed short-time spectra Sd[row_b] + sum_g gain[b][g] STFT(tau_g): G STFTs per '
                                     'band and step, none per receiver' if (fused.linear_transforms and fused.spectral_edr)
                                     else 'per-receiver STFT of the stored time signals')),
                       'dataset_constants': ('precomputed once per dataset before the timed region, like the rFFT front end '
                                             '(SURVEY section 8d): target EDR / EDC; xd = irfft(early response x band filter) '
                                             'per receiver; Sd = STFT(xd) per receiver -- parameter-independent linear images '
                                             'of the dataset'),
                       'launch': graph_mode, 'rank_devices': rank_devices,
                       'collective': (None if world == 1 else
                                      {'backend': dist.get_backend(), 'library_version': collective_version(),
                                       'per_step': 'ONE all-reduce (sum) of [gradients of all bands | 3 loss slots per band]',
                                       'floats': int(trainer.optimizer.bucket.numel())}),
                       'final_loss': [float(v) for v in torch.as_tensor(total).reshape(-1).tolist()]},
        }
        if extra:
            out['extra'] = extra
        if epoch_info is not None:
            out['epoch'] = epoch_info
        if ktimes:
            named = ktimes['kernel']
            units = ktimes['units_per_launch']
            per = ALG_BYTES_PER_UNIT[named]
            in_step_us = ktimes['avg_ms'] * 1e3            # the raw bracket: nothing subtracted
            achieved = units * per / (in_step_us * 1e-6) / 1e9
            pair = None
            if 'window' in kpair:
                alg = sum(kpair[k]['units_per_launch'] * ALG_BYTES_PER_UNIT[k] for k in PAIR_KERNELS if k in kpair)
                trf = [pmc_traffic_bytes(k) for k in PAIR_KERNELS]
                wus = kpair['window']['avg_ms'] * 1e3
                pair = {'kernels': {k: {'in_step_us': kpair[k]['avg_ms'] * 1e3,
                                        'isolated_us': (iso_pair.get(k, {}).get('avg_ms', float('nan')) * 1e3),
                                        'alg_bytes_per_launch': kpair[k]['units_per_launch'] * ALG_BYTES_PER_UNIT[k],
                                        'traffic': pmc_traffic_bytes(k)} for k in PAIR_KERNELS if k in kpair},
                        'window_us': wus, 'alg_bytes': alg, 'achieved_GBs': alg / wus / 1e3,
                        'frac': alg / wus / 1e3 / HBM_PEAK_GBS,
                        'window_us_in_replay': replayed_pair_window(),
                        'frac_in_replay': ((alg / replayed_pair_window() / 1e3 / HBM_PEAK_GBS)
                                           if replayed_pair_window() else None),
                        'traffic': (sum(trf) if all(t is not None for t in trf) else None),
                        'traffic_frac': ((sum(trf) / wus / 1e3 / HBM_PEAK_GBS) if all(t is not None for t in trf) else None),
                        'what': 'the two memory-heavy launches of the step run beside each other (main / side stream): window '
                                '= first start to last end of the pair, HIP events, averaged over the host-launched steps'}
            out['roofline'] = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': achieved / HBM_PEAK_GBS, 'traffic': pmc_traffic_bytes(named),
                               'kernel': named, 'avg_launch_us': in_step_us, 'pair': pair,
                               'kernel_chosen': 'the longer of the pair (k_edr_lin_wave, k_edc_lin_one) in the REPLAYED step '
                                                f'(profiles/{PROFILE_TAG}_step_kernel_durations.csv); both are bracketed live',
                               'measured': 'HIP events on the launch stream around every launch of the kernel during '
                                           f'{ROOFLINE_EAGER_STEPS} steps of the timed launch sequence (host launches, '
                                           'same streams and concurrency as the replayed graph); raw bracket',
                               'event_pair_overhead_us': ktimes['event_pair_overhead_ms'] * 1e3,
                               'isolated_us': iso.get('avg_ms', float('nan')) * 1e3,
                               'isolated_frac': (units * per / (iso['avg_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS) if iso else None,
                               'launches': ktimes['launches'], 'alg_bytes_per_unit': per,
                               'alg_bytes_per_launch': units * per, 'units_per_launch': units,
                               'top': roofline_top(int(round(units)),
                                                   signals=(nbands * G if getattr(fused, 'linear_transforms', False) else None))}
        if not sub_record:
            out['build'] = library_stamp()
        out['whole_step_hbm_frac'] = rirs_per_s / world * ALG_BYTES_PER_RIR / 1e9 / HBM_PEAK_GBS
        out['whole_step_alg_GBs'] = rirs_per_s / world * ALG_BYTES_PER_RIR / 1e9
        # what the step really moves: the bytes the linear step has to move per RIR (LINEAR_ALG_BYTES_PER_RIR) and, measured,
        # the PMC traffic of the replayed step's launches (profiles/<tag>_step_traffic.json, tools/make_profiles.py)
        out['linear_alg_bytes_per_rir'] = LINEAR_ALG_BYTES_PER_RIR
        out['linear_alg_hbm_frac'] = rirs_per_s / world * LINEAR_ALG_BYTES_PER_RIR / 1e9 / HBM_PEAK_GBS
        st = step_traffic()
        if st is not None and use_bank and nbands == len(BAND_CENTRES) and NPER == 4 and not sub_record:
            out['step_traffic_bytes'] = st['bytes_per_step']
            out['step_traffic_frac'] = st['bytes_per_step'] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
            out['step_traffic_source'] = st['source']
        if world == 1 and not args.no_cpu_baseline and not sub_record:
            out['cpu_baseline'] = cpu_baseline(room, delays, filt.cpu().numpy().astype(np.complex128), device=device,
                                               steps=args.cpu_steps)
    # ---- the other two BASELINE.json configurations on the same clock (N = 1, default run): short sub-records
    if world == 1 and not sub_record and not args.no_extras and not args.epoch and use_bank and nbands == len(BAND_CENTRES) \
            and NPER == 4 and RECIPE is None:
        del step, trainer, data, net
        torch.cuda.empty_cache()
        out.setdefault('extra', {})
        out['extra']['n32'] = sub_bench(args, device, 'n32')
        torch.cuda.empty_cache()
        out['extra']['reference_recipe'] = sub_bench(args, device, 'recipe')
        torch.cuda.empty_cache()
        out['extra']['directional'] = sub_bench(args, device, 'directional')
    return out


def sub_bench(args, device, which):
    """BASELINE.json configs[4] (N = 32 = 4 x 8 lines) / configs[3] (directional) as short sub-records of the default line, so
    that the driver's clock covers them too: ``--extra-steps`` timed steps after 5 warm-up steps, no roofline / CPU legs
    (those are in ``bench.py --lines-per-group 8`` / ``--config directional``)."""
    import copy
    a = copy.copy(args)
    a.steps, a.warmup, a.no_cpu_baseline = args.extra_steps, 5, True
    t0 = time.perf_counter()
    if which == 'n32':
        globals()['NPER'] = 8
        try:
            r = run_omni(a, device, 0, 1, [0], [0], sub_record=True)
        finally:
            globals()['NPER'] = 4
        keep = ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'whole_step_hbm_frac')
        rec = {k: r[k] for k in keep}
        rec['config'] = {k: r['config'][k] for k in ('workload', 'delay_lines', 'rirs_per_s', 'final_loss')}
    elif which == 'recipe':
        saved = (RECIPE, G, BAND_CENTRES)
        globals().update(RECIPE='reference', G=3, BAND_CENTRES=REFERENCE_RECIPE_CENTRES)
        a.bands = len(REFERENCE_RECIPE_CENTRES)
        try:
            r = run_omni(a, device, 0, 1, [0], [0], sub_record=True)
        finally:
            globals().update(RECIPE=saved[0], G=saved[1], BAND_CENTRES=saved[2])
        rec = {k: r[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'whole_step_hbm_frac')}
        rec['config'] = {k: r['config'][k] for k in ('workload', 'delay_lines', 'rirs_per_s', 'final_loss')}
        rec['config']['gain_networks'] = {str(int(f)): '%d x %d' % recipe_network_of('reference', f)
                                          for f in REFERENCE_RECIPE_CENTRES}
        rec['config']['ms_per_band_step'] = r['ms_per_step'] / len(REFERENCE_RECIPE_CENTRES)
    else:
        r = run_directional(a, device, 0, 1)
        rec = {k: r[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup')}
        rec['config'] = {k: r['config'][k] for k in ('workload', 'delay_lines', 'rirs_per_s', 'ms_per_band_step', 'final_loss')}
        if 'roofline' in r:
            rec['roofline'] = r['roofline']
    rec['wall_s_incl_build'] = time.perf_counter() - t0
    return rec


if __name__ == '__main__':
    main()
