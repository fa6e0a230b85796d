#!/usr/bin/env python
"""Benchmark of the DiffGFDN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (the configuration BASELINE.json's metric is quoted on -- "N=16 delay lines, 838 pos x 7
bands" -- which fits one GPU): 7 octave-band GFDNs (63 Hz ... 4 kHz), each N = 16 delay lines
(4 groups x 4) on its own 838-receiver synthetic grid, nfft = 131 072 (K = 65 537 bins), fs = 32 kHz,
batch of 32 receiver positions per band per optimiser step, losses EDR(w=1) + EDC(w=10, random mask)
+ asymmetric spectral(w=1) + sparsity(w=2), Adam -- the recipe of the reference's
src/run_subband_training_treble.py:105-154, which trains the bands one after another; here the band
bank (diffgfdn_amd/bandbank.py) steps all of them with one launch per stage.  ``--bands 1`` runs
BASELINE.json configs[1] alone (the 500 Hz band).

One "step" = what the reference's training loop does per batch (trainer.py:373-379), for every band:
normalize (no-grad sub-FDN forward + in-place rescale of b, c) + train_step (forward, losses,
backward, [all-reduce], Adam), on 32 receivers per band drawn from the band's grid.  Inputs are
resident in HBM before the timed region.  Metric: RIR-frames/s = bands x receivers x 32 EDR frames /
second, summed over ranks (weak scaling: every rank steps its own 32-receiver shard per band of a
32N global batch and the parameter gradients are summed by one flat RCCL all-reduce).

The JSON line also carries
  roofline     : HBM roofline of the dominant kernel, timed live with HIP events on the launch
                 stream inside the timed region;
  cpu_baseline : the same step on the host cores by the CPU oracle (oracle/cpu_trainer.py, a
                 restatement pinned to the reference; kind "port"), rank 0 at N = 1 only.
"""
import argparse
import json
import os
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
NUM_RECEIVERS = 838
G, NPER = 4, 4
BATCH = 32
BAND_CENTRES = (63.0, 125.0, 250.0, 500.0, 1000.0, 2000.0, 4000.0)
FRAMES = 32
WIN = 4096
HBM_PEAK_GBS = 8000.0
# compulsory HBM bytes per RIR (fwd + bwd, fp32 / complex64 storage)
ALG_BYTES_PER_RIR = 2811048
# Dominant kernel by total time in profiles/r01_bench_kernel_stats.csv: the first column pass of the
# odd-length irfft (k_blu_col128_fwd; input side + FFT over n1 + twiddle), HBM-bound, one forward and one
# adjoint launch per step (the roofline leg averages both).  Two items share one transform, so per RIR
# (DESIGN.md §kernels): the forward launch reads the item's slot-ordered spectrum (8 B x 32769) and writes half
# a work block (8 B x 2^16 / 2) = 524 296 B; the adjoint launch gathers ONE real input (4 B x 65537: the STFT
# adjoint already added the EDC gradient) and writes half a work block = 524 292 B.  Average: 524 294 B.
DOMINANT_KERNEL = 'k_blu_col128_fwd'
DOMINANT_ALG_BYTES_PER_UNIT = (8 * 32769 + 4 * 65536 + 4 * 65537 + 4 * 65536) // 2
# HBM traffic per launch from rocprofv3 --pmc passes (profiles/r01_pmc_hbm_bytes.csv), by items per launch:
# 2 x FETCH_SIZE (gfx950) + WRITE_SIZE, averaged over the forward and the adjoint launch like the duration
DOMINANT_TRAFFIC_BYTES_PER_LAUNCH = {224: int(2 * 30323.8 * 1024 + 57400.2 * 1024)}
ROOFLINE_EAGER_STEPS = 20
CPU_BASELINE_THREADS = 16            # the torch CPU path anti-scales beyond this on the 2x64-core host


def octave_band_response(centre_hz: float, fs: float, nfft: int, numtaps: int = 2049) -> np.ndarray:
    """(K,) response of a linear-phase octave band-pass FIR (stands in for the pyfar taps the
    reference loads, trainer.py:116-128 -- the taps are input data on this path)."""
    from scipy.signal import firwin
    lo, hi = centre_hz / np.sqrt(2.0), centre_hz * np.sqrt(2.0)
    taps = firwin(numtaps, [lo, hi], pass_zero=False, fs=fs)
    return np.fft.rfft(taps, n=nfft)


def build_workload(device, seed: int, num_receivers: int = NUM_RECEIVERS, centre_hz: float = 500.0,
                   room_seed: int = 0, make_trainer: bool = True):
    from diffgfdn_amd.config import (CouplingMatrixType, DiffGFDNConfig, FeedbackLoopConfig,
                                     OutputFilterConfig, SubbandProcessingConfig, TrainerConfig)
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset, split_dataset
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    from diffgfdn_amd.synthetic import synthetic_room
    from diffgfdn_amd.trainer import VarReceiverPosTrainer

    room = synthetic_room(num_receivers, G, FS, 64000, seed=room_seed)
    ds = RoomDataset(G, FS, room['source_position'], room['receiver_position'], room['rirs'],
                     room['common_decay_times'], nfft=NFFT, device=device)
    data = MultiRIRDataset(device, ds)
    torch.manual_seed(seed)
    np.random.seed(seed)
    cfg = DiffGFDNConfig(num_groups=G, num_delay_lines=G * NPER, sample_rate=FS, seed=23463 + int(centre_hz))
    delays = cfg.delay_length_samps
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=5, num_neurons_per_layer=16,
                            num_fourier_features=20)
    net = DiffGFDNVarReceiverPos(FS, G, delays, device, fl, of, use_absorption_filters=False,
                                 common_decay_times=room['common_decay_times'],
                                 use_colorless_loss=True).to(device)
    tc = TrainerConfig(batch_size=BATCH, num_freq_bins=NFFT, max_epochs=20, lr=1e-3, io_lr=1e-2,
                       use_edc_mask=True, use_colorless_loss=True, edc_loss_weight=10,
                       sparsity_loss_weight=2, use_asym_spectral_loss=True, device='cuda',
                       train_dir='/tmp/gfdn_bench/train', ir_dir='/tmp/gfdn_bench/ir',
                       subband_process_config=SubbandProcessingConfig(
                           centre_frequency=centre_hz, frequency_range=(63, 8000), num_fraction_octaves=1))
    filt = torch.tensor(octave_band_response(centre_hz, FS, NFFT), device=device).to(torch.complex64)
    train_idx, valid_idx, test_idx = split_dataset(data, 0.8, test_ratio=0.1)
    if not make_trainer:
        return room, data, net, tc, train_idx, filt, delays
    pg = dist.group.WORLD if dist.is_initialized() else None
    trainer = VarReceiverPosTrainer(net, tc, subband_filter_freq_resp=filt, process_group=pg,
                                    capturable=True)
    start, length = trainer.criterion[1].window(K)
    data.precompute_decay_targets(WIN, start, length)
    return room, data, net, trainer, train_idx, filt, delays


def build_bank_workload(device, seed: int, centres, num_receivers: int = NUM_RECEIVERS):
    """One model + dataset per octave band (run_subband_training_treble.py:175-204), stacked into a
    band bank.  Returns the 500 Hz band's room / delays / filter for the CPU baseline leg."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    nets, datasets, filts, splits = [], [], [], []
    cpu_leg = None
    for q, f in enumerate(centres):
        room, data, net, tc, train_idx, filt, delays = build_workload(
            device, seed + q, num_receivers, centre_hz=f, room_seed=q, make_trainer=False)
        nets.append(net)
        datasets.append(data)
        filts.append(filt)
        splits.append(train_idx)
        if cpu_leg is None or f == 500.0:          # the CPU baseline leg times the 500 Hz band
            cpu_leg = (room, delays, filt)
    bank = BandBank(nets)
    pg = dist.group.WORLD if dist.is_initialized() else None
    trainer = BandBankTrainer(bank, tc, subband_filter_freq_resp=torch.stack(filts), process_group=pg,
                              band_names=[int(f) for f in centres])
    sds = BandStackedDataset(datasets, free_sources=True)
    start, length = trainer._decay_window(K)
    sds.precompute_decay_targets(WIN, start, length)
    for d in datasets:                       # only the stacked stores are read from here on
        d.rir_mag_response = d.late_rir_mag_response = None
    sds.rir_mag_response = torch.empty((1, K), dtype=torch.complex64, device=device)   # shape carrier
    torch.cuda.empty_cache()
    return cpu_leg, sds, bank, trainer, splits


def hip_losses_at(init, batch, delays, room, filt_np, device):
    """Loss terms of the HIP path at the parameters / batch the CPU baseline starts from (normalize, then one
    forward + losses, no step): the other side of ``loss_delta_vs_cpu``."""
    from diffgfdn_amd.config import (CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig,
                                     SubbandProcessingConfig, TrainerConfig)
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=5, num_neurons_per_layer=16, num_fourier_features=20)
    net = DiffGFDNVarReceiverPos(FS, G, delays, device, fl, of, use_absorption_filters=False,
                                 common_decay_times=room['common_decay_times'], use_colorless_loss=True).to(device)
    with torch.no_grad():
        net.input_gains.copy_(init['input_gains'])
        net.output_gains.copy_(init['output_gains'])
        net.feedback_loop.M.copy_(init['M'])
    net.output_scalars.mlp.load_state_dict(init['mlp'])
    tc = TrainerConfig(batch_size=BATCH, num_freq_bins=NFFT, max_epochs=1, lr=1e-3, io_lr=1e-2, use_edc_mask=False,
                       use_colorless_loss=True, edc_loss_weight=10, sparsity_loss_weight=2,
                       use_asym_spectral_loss=True, device='cuda', train_dir='/tmp/gfdn_bench/train',
                       ir_dir='/tmp/gfdn_bench/ir',
                       subband_process_config=SubbandProcessingConfig(centre_frequency=500.0, frequency_range=(63, 8000),
                                                                      num_fraction_octaves=1))
    tr = VarReceiverPosTrainer(net, tc, subband_filter_freq_resp=torch.tensor(filt_np, device=device).to(torch.complex64))
    b = {k: v.to(device) for k, v in batch.items()}
    with torch.no_grad():
        tr.normalize(b)
        out = tr._step_losses(b, draw_mask=False)
    torch.cuda.synchronize()
    return {k: float(v) for k, v in out.items() if k.endswith('_loss')}


def cpu_baseline(room, delays, filt_np, steps: int = 2, device=None):
    """The same optimiser step on the host cores with the CPU oracle (reference restatement); with ``device`` also
    the loss terms of the HIP path on the same parameters and batch (``loss_delta_vs_cpu``, the metric's
    "EDR loss delta vs ref")."""
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
    lin = [(m.weight.detach().clone(), m.bias.detach().clone()) for m in mlp.model if isinstance(m, torch.nn.Linear)]
    norm = [(m.weight.detach().clone(), m.bias.detach().clone()) for m in mlp.model if isinstance(m, torch.nn.LayerNorm)]
    N = G * NPER
    p = orc.GridModelParams(FS, delays, G, (2 * torch.randn(N, 1) - 1) / N, (2 * torch.randn(N, 1) - 1) / N,
                            (2 * torch.rand(G, NPER, NPER) - 1) / np.sqrt(NPER), torch.zeros(G * (G - 1) // 2),
                            room['common_decay_times'], lin, norm, 20)
    init = {'input_gains': p.input_gains.detach().clone(), 'output_gains': p.output_gains.detach().clone(),
            'M': p.M.detach().clone(), 'mlp': {k: v.detach().clone() for k, v in mlp.state_dict().items()}}
    tr = OracleGridTrainer(p, lr=1e-3, io_lr=1e-2, edr_weight=1.0, edc_weight=10.0, spectral_weight=1.0,
                           sparsity_weight=2.0, use_asym=True, subband_filter=torch.tensor(filt_np))
    L = min(orc.ms_to_samps(float(np.max(room['common_decay_times'])) * 1e3, FS), K) - mix
    times, first = [], None
    for s in range(steps + 1):           # first step is warm-up (allocator, thread pools)
        mask = torch.argwhere(torch.bernoulli(torch.empty(L).uniform_(0, 1)))
        t0 = time.time()
        tr.normalize(batch)
        _, terms = tr.train_step(batch, mask)
        times.append(time.time() - t0)
        first = terms if first is None else first      # loss terms at the initial (normalized) parameters
    sec = float(np.mean(times[1:]))
    out = {'value': BATCH * FRAMES / sec, 'unit': 'RIR-frames/s', 'cores': cores, 'kind': 'port',
           'sample': f'{steps} optimiser steps (normalize + fwd + EDR/EDC/colorless losses + bwd + Adam) of the '
                     f'same workload, batch {BATCH}, after 1 warm-up step; {sec:.2f} s/step',
           'sec_per_step': sec}
    if device is not None:
        # mask-free terms only (the EDC term of the CPU step used a random time mask)
        ours = hip_losses_at(init, batch, delays, room, filt_np, device)
        out['loss_delta_vs_cpu'] = {k: {'cpu': first[k], 'hip': ours[k], 'rel': abs(ours[k] - first[k]) / abs(first[k])}
                                    for k in ('edr_loss', 'spectral_loss', 'sparsity_loss')}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='launch every kernel from the host (no HIP graph)')
    ap.add_argument('--cpu-steps', type=int, default=2)
    ap.add_argument('--receivers', type=int, default=NUM_RECEIVERS)
    ap.add_argument('--classic', action='store_true',
                    help='with --bands 1: the per-band VarReceiverPosTrainer path instead of a one-band bank')
    ap.add_argument('--lines-per-group', type=int, default=NPER,
                    help='delay lines per group (4 groups): 8 gives the N = 32 configuration (with --classic --bands 1)')
    ap.add_argument('--bands', type=int, default=len(BAND_CENTRES),
                    help='octave bands stepped together (1 = BASELINE.json configs[1], the 500 Hz band alone)')
    args = ap.parse_args()
    if args.lines_per_group != NPER:
        if not (args.classic or args.lines_per_group <= 4):
            ap.error('--lines-per-group > 4 needs --classic --bands 1 (the band bank fuses 4 x 4 blocks)')
        globals()['NPER'] = args.lines_per_group
    if not 1 <= args.bands <= len(BAND_CENTRES):
        raise SystemExit(f"--bands must be 1..{len(BAND_CENTRES)}")
    if args.classic and args.bands != 1:
        raise SystemExit("--classic steps one band: use --bands 1")

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    # host-side torch ops on tiny CPU tensors (mask draw, index lists) crawl when the intra-op pool
    # spans all 256 host cores (measured 30 ms/step at 128 threads vs 6 ms at 4)
    torch.set_num_threads(min(4, os.cpu_count() or 1))
    # Rehearsal knobs for a ONE-GPU box (the driver's multi-GPU run uses neither): GFDN_BENCH_ONE_DEVICE=1 puts every
    # rank on cuda:0 and GFDN_BENCH_BACKEND=gloo carries the collectives (RCCL refuses two ranks on one device) --
    # the N > 1 code path (shared mask seed, split graphs, flat-gradient all-reduce, max-over-ranks timing) end to end.
    dev_index = 0 if os.environ.get('GFDN_BENCH_ONE_DEVICE') else local_rank
    backend = os.environ.get('GFDN_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    from diffgfdn_amd import hip_ops
    nbands = args.bands
    use_bank = not args.classic
    if not use_bank:
        room, data, net, trainer, train_idx, filt, delays = build_workload(device, seed=1234,
                                                                          num_receivers=args.receivers)
        splits = [train_idx]
        centres = (500.0,)
    else:
        centres = BAND_CENTRES[:nbands] if nbands > 1 else (500.0,)
        (room, delays, filt), data, net, trainer, splits = build_bank_workload(device, 1234, centres,
                                                                               args.receivers)
    # every rank draws its own receivers (different shards of a 32*world global batch per band)
    gen = torch.Generator().manual_seed(100 + rank)
    splits_t = [torch.tensor(s) for s in splits]

    def draw():
        sel = [t[torch.randperm(len(t), generator=gen)[:BATCH]].tolist() for t in splits_t]
        return sel[0] if not use_bank else data.global_rows(sel)

    step = trainer.graphed(data, BATCH)      # normalize + train_step as HIP-graph replays

    def one_step():
        sel = draw()
        if args.eager:
            batch = data.collate(sel, lean=True) if not use_bank else data.collate(sel)
            trainer.normalize(batch)
            return trainer.train_step(batch)
        losses = step(sel)
        return losses['_total'], losses

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total, parts = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # roofline leg: the graph replays above cannot carry per-kernel events, so the SAME step is run
    # ROOFLINE_EAGER_STEPS more times with host launches and the dominant kernel bracketed by HIP
    # events on the launch stream (same kernels, same shapes, same data)
    ktimes = {}
    if rank == 0:
        hip_ops.kernel_timer.watch = DOMINANT_KERNEL
        hip_ops.kernel_timer.start()
        ones = torch.ones(() if nbands == 1 else (nbands,), device=device)
        for _ in range(ROOFLINE_EAGER_STEPS):
            sel = draw()
            batch = data.collate(sel, lean=True) if not use_bank else data.collate(sel)
            with torch.no_grad():
                trainer.normalize(batch)
            # no optimizer step / all-reduce (no collectives inside): kernel timing only
            if not use_bank:
                lo = trainer._step_losses(batch, mask_prenorm=step.maskw)
                lo['_total'].backward()
            else:
                lo = trainer._step_losses(batch, mask_prenorm=step.maskw, defer_total=True)
                torch.autograd.backward(lo['_heads'], [ones, ones])
        ktimes = hip_ops.kernel_timer.stop()
        trainer.optimizer.zero_grad(set_to_none=True)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    rirs_per_s = nbands * BATCH * world * args.steps / elapsed
    value = rirs_per_s * FRAMES

    if rank == 0:
        out = {
            'metric': 'RIR-frames/sec', 'value': value, 'unit': 'RIR-frames/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': f'{nbands} octave-band GFDN(s) ({", ".join(str(int(f)) for f in centres)} Hz), each '
                                   f'N=16 (4 groups x 4) on a {args.receivers}-receiver grid, nfft 131072 (K=65537), '
                                   'batch 32 receivers/band/step/GPU; step = normalize + fwd + EDR/EDC(mask)/'
                                   'colorless losses + bwd + Adam for every band'
                                   + (' (band bank: one launch per stage for all bands)' if use_bank else ''),
                       'bands': nbands, 'receivers': args.receivers, 'batch_per_band_per_gpu': BATCH,
                       'rirs_per_step_per_gpu': nbands * BATCH, 'global_batch_per_band': BATCH * world,
                       'delay_lines': G * NPER, 'bins': K, 'rirs_per_s': rirs_per_s,
                       'launch': 'eager' if args.eager else 'hip_graph',
                       'final_loss': [float(v) for v in total.reshape(-1).tolist()]},
        }
        dom = ktimes if ktimes else None
        if dom:
            dom['alg_bytes_per_unit'] = DOMINANT_ALG_BYTES_PER_UNIT
            units = dom['units_per_launch']
            # launch duration = bracket (start event .. end event) minus what an EMPTY event pair
            # measures on the same stream; rocprofv3's average for the kernel is the cross-check
            # (profiles/r01_bench_kernel_stats.csv)
            net_ms = max(dom['avg_ms'] - dom['event_pair_overhead_ms'], 1e-6)
            achieved = units * dom['alg_bytes_per_unit'] / (net_ms * 1e-3) / 1e9
            out['roofline'] = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': achieved / HBM_PEAK_GBS,
                               'traffic': DOMINANT_TRAFFIC_BYTES_PER_LAUNCH.get(int(round(units))),
                               'kernel': dom['kernel'], 'avg_launch_us': net_ms * 1e3,
                               'bracket_us': dom['avg_ms'] * 1e3,
                               'event_pair_overhead_us': dom['event_pair_overhead_ms'] * 1e3,
                               'launches': dom['launches'],
                               'alg_bytes_per_launch': units * dom['alg_bytes_per_unit']}
        out['whole_step_hbm_frac'] = rirs_per_s / world * ALG_BYTES_PER_RIR / 1e9 / HBM_PEAK_GBS
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(room, delays, filt.cpu().numpy().astype(np.complex128), device=device,
                                               steps=args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                 # rank 0 may still be in its roofline / baseline legs
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
