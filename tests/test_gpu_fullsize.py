"""Full-size (nfft = 131 072, K = 65 537, N = 16) BACKWARD parity against the CPU oracle.

The production configuration runs kernels that exist only at this size (Rader on 2^16 points, slot-ordered spectra,
two items per transform, the 4096-point register FFTs), so the gradients -- not only the forward values -- are
pinned to the oracle here, with the device's Philox time mask handed to the oracle as the reference's index set
(losses.py:221-227):
  (i)  VarReceiverPosTrainer (per-bin elimination kernels through autograd), batch 2;
  (ii) a 2-band BandBankTrainer on the graph-replayed explicit step (polynomial-form block transfer functions,
       slot order, pair-interleaved signals), batch 2 per band.
Compared: every weighted loss term (1e-4 relative, the north star's bar), every parameter gradient (max-norm per
tensor, 2e-3), and the parameters after the Adam update."""
import os

import numpy as np
import pytest
from tests.margins import within
import torch

from oracle import gfdn_oracle as orc
from oracle.cpu_trainer import OracleGridTrainer
from tests.helpers import philox_mask, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
FS, NFFT, G, NPER, R, B = 32000.0, 131072, 4, 4, 6, 2
K = NFFT // 2 + 1
CENTRES = (250.0, 1000.0)
DELAYS = [[641, 701, 809, 907, 1009, 1103, 1201, 1301, 1399, 1409, 1423, 1427, 1429, 1433, 1439, 1601],
          [643, 709, 811, 911, 1013, 1109, 1213, 1303, 1381, 1411, 1423, 1427, 1447, 1451, 1453, 1601]]
LOSS_TOL, GRAD_TOL = 1e-4, 2e-3
# dL/dM of the timed bench shape per lines-per-group: what is measured (round 4: 3.6e-4 at N = 16, 7.9e-4 at N = 32, the float32
# transforms upstream of the records are the floor -- DESIGN.md section 2 (iv), profiles/r04_grad_stage_probe.txt) plus margin
GRAD_TOL_M = {4: 2.7e-4, 8: 4.5e-4}        # (measured worst band 1.8e-4 / 2.9e-4 with the float64 direct-path store + 50 %)
# BASELINE.json configs[4]: N = 32 = 4 groups x 8 lines (mutually prime delays, as DiffGFDNConfig draws them)
DELAYS32 = [571, 593, 613, 631, 653, 673, 691, 709, 733, 751, 769, 787, 809, 827, 853, 877, 907, 929, 947, 967, 983, 1009,
            1031, 1051, 1069, 1091, 1109, 1129, 1151, 1171, 1193, 1213]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.fixture
def float32_direct_store():
    """the linear step's direct-path store built with the float32 transforms of rounds 4-5: what a test that compares the
    linear step with the stored-signal step needs (both then see the SAME float32 direct paths; the float64 store of round 6
    is closer to the reference by the float32 transform's error, 4e-6 on a loss)"""
    from diffgfdn_amd.bandbank import BandStackedDataset
    BandStackedDataset.direct_time_f64 = False
    yield
    BandStackedDataset.direct_time_f64 = True



def _filters():
    from scipy.signal import firwin
    return np.stack([np.fft.rfft(firwin(1025, [f / np.sqrt(2), f * np.sqrt(2)], pass_zero=False, fs=FS), n=NFFT)
                     for f in CENTRES])


def _band(q, delays=None, t60max=1.5):
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    from diffgfdn_amd.synthetic import synthetic_room
    room = synthetic_room(R, G, FS, 40000, seed=40 + q, t60_range=(0.3, t60max))
    ds = MultiRIRDataset(DEV, RoomDataset(G, FS, room["source_position"], room["receiver_position"], room["rirs"],
                                          room["common_decay_times"], nfft=NFFT, device=DEV))
    torch.manual_seed(200 + q)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    net = DiffGFDNVarReceiverPos(FS, G, DELAYS[q] if delays is None else delays, DEV, fl, of, use_absorption_filters=False,
                                 common_decay_times=room["common_decay_times"], use_colorless_loss=True).to(DEV)
    return room, ds, net


def _tc():
    from diffgfdn_amd.config import SubbandProcessingConfig, TrainerConfig
    return TrainerConfig(batch_size=B, num_freq_bins=NFFT, lr=1e-3, io_lr=1e-2, use_colorless_loss=True,
                         use_asym_spectral_loss=True, edc_loss_weight=10.0, sparsity_loss_weight=2.0,
                         use_edc_mask=True, train_dir="/tmp/gfdn_full/t", ir_dir="/tmp/gfdn_full/a", device="cuda",
                         subband_process_config=SubbandProcessingConfig(centre_frequency=500.0,
                                                                        frequency_range=(63, 8000),
                                                                        num_fraction_octaves=1))


def _oracle_job(sd, q, room, ds, sel, filt_q, keep, delays=None, n_fourier=4):
    """what tests/oracle_jobs.grid_step needs for normalize + train_step of the CPU oracle from the state dict ``sd`` on
    receivers ``sel`` (CPU tensors only)"""
    idx = torch.tensor(sel)
    ob = {"z_values": ds.z_values.cpu(),
          "norm_listener_position": ds.norm_listener_position[idx].cpu(),
          "listener_position": ds.listener_positions[idx].cpu(),
          "target_early_response": ds.early_response_c128(idx).cpu(),
          "target_rir_response": ds.rir_mag_response[idx].cpu().to(torch.complex128)}
    return {"fs": FS, "delays": DELAYS[q] if delays is None else delays, "G": G,
            "sd": {k: v.detach().cpu().clone() for k, v in sd.items()}, "common_decay_times": room["common_decay_times"],
            "n_fourier": n_fourier, "filt": filt_q.cpu().to(torch.complex128), "batch": ob, "keep": keep, "threads": 16}


def _oracle_step(sd, q, room, ds, sel, filt_q, keep, delays=None, n_fourier=4):
    """normalize + train_step of the CPU oracle from the state dict ``sd`` on receivers ``sel``; returns the loss
    parts, the gradients and the parameters after Adam, keyed like the model's state dict."""
    from tests.oracle_jobs import grid_step
    return grid_step(_oracle_job(sd, q, room, ds, sel, filt_q, keep, delays, n_fourier))


def _check(tag, parts_hip, grads_hip, after_hip, before, parts, grads, after, grad_tol=None, grad_tol_M=None):
    grad_tol = GRAD_TOL if grad_tol is None else grad_tol
    for k, v in parts.items():
        assert abs(parts_hip[k] - v) <= LOSS_TOL * abs(v) + 1e-7, (tag, k, parts_hip[k], v)
    worst = {}
    for k, g in grads.items():
        gh = np.asarray(grads_hip[k], dtype=np.float64).reshape(-1)
        go = g.numpy().astype(np.float64).reshape(-1)
        worst[k] = np.abs(gh - go).max() / (np.abs(go).max() + 1e-300)
        if k == "feedback_loop.M":
            print(f"[parity] {tag}: dL/dM max deviation {worst[k]:.3e} of its largest entry")
        assert worst[k] < (grad_tol_M if (grad_tol_M is not None and k == "feedback_loop.M") else grad_tol), \
            (tag, k, worst[k])
        # Adam's first step moves every entry by lr g / (|g| + eps): the update of M (which normalize leaves alone) is
        # compared where the gradient is not at the noise floor of float32 sums over 65 537 bins
        if k == "feedback_loop.M":
            b0 = np.asarray(before[k], dtype=np.float64).reshape(-1)
            d_hip = np.asarray(after_hip[k], dtype=np.float64).reshape(-1) - b0
            d_ora = after[k].numpy().astype(np.float64).reshape(-1) - b0
            big = np.abs(go) > 1e-3 * np.abs(go).max()
            within(np.abs(d_hip[big] - d_ora[big]).max() / np.abs(d_ora[big]).max(), 1e-4, ("fullsize L129", tag, k))
    return worst


def _mask(seed, step, length, gb):
    mw_np, _ = philox_mask(seed, step, length, 1.0 / gb)
    return torch.tensor(mw_np, device=DEV), torch.argwhere(torch.tensor(mw_np) > 0)


def test_full_size_single_band_backward_vs_oracle():
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    room, ds, net = _band(0)
    filt = torch.tensor(_filters()[0], device=DEV).to(torch.complex64)
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    tr = VarReceiverPosTrainer(net, _tc(), subband_filter_freq_resp=filt, capturable=True)
    start, length = tr._decay_window(K)
    mw, keep = _mask(4242, 0, length, B)
    sel = [1, 4]
    batch = ds.collate(sel)
    tr.normalize(batch)
    tr.optimizer.zero_grad(set_to_none=True)
    losses = tr._step_losses(batch, mask_prenorm=mw)
    losses.pop("_total").backward()
    grads_hip = {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters() if p.grad is not None}
    tr.optimizer.step()
    parts_hip = {k: float(v) for k, v in losses.items() if k.endswith("_loss")}
    after_hip = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    parts, grads, after = _oracle_step(sd0, 0, room, ds, sel, filt, keep)
    worst = _check("single", parts_hip, grads_hip, after_hip, {k: v.numpy() for k, v in sd0.items()}, parts, grads,
                   after)
    for name in ("input_gains", "output_gains"):         # normalize + Adam: the full update of the gains
        assert rel_err(after_hip[name], after[name].numpy()) < 1e-4, name
    print("single-band full-size gradient deviations:", {k: f"{v:.1e}" for k, v in worst.items()})


def test_full_size_n32_backward_vs_oracle():
    """BASELINE.json configs[4] (N = 32 = 4 groups x 8 lines) at K = 65 537: losses, every gradient and the Adam update of
    the module on the thread-per-system 8 x 8 elimination kernels (k_solve8_fwd / _bwd, k_subfdn8_energy) against the
    oracle -- the float32 leg of the configuration whose bf16 leg DESIGN section 8 measures and rejects."""
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    room, ds, net = _band(0, DELAYS32)
    assert net.num_delay_lines_per_group == 8
    filt = torch.tensor(_filters()[0], device=DEV).to(torch.complex64)
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    tr = VarReceiverPosTrainer(net, _tc(), subband_filter_freq_resp=filt, capturable=True)
    start, length = tr._decay_window(K)
    mw, keep = _mask(99, 0, length, B)
    sel = [0, 5]
    batch = ds.collate(sel)
    tr.normalize(batch)
    tr.optimizer.zero_grad(set_to_none=True)
    losses = tr._step_losses(batch, mask_prenorm=mw)
    losses.pop("_total").backward()
    grads_hip = {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters() if p.grad is not None}
    tr.optimizer.step()
    parts_hip = {k: float(v) for k, v in losses.items() if k.endswith("_loss")}
    after_hip = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    parts, grads, after = _oracle_step(sd0, 0, room, ds, sel, filt, keep, DELAYS32)
    worst = _check("n32", parts_hip, grads_hip, after_hip, {k: v.numpy() for k, v in sd0.items()}, parts, grads, after)
    print("N = 32 full-size gradient deviations:", {k: f"{v:.1e}" for k, v in worst.items()})


def test_full_size_bank_graph_step_backward_vs_oracle():
    """The path bench.py times: 2 bands, graph replay of the explicit step (slot order, pairs), device-drawn mask."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    bands = [_band(q) for q in range(2)]
    filt = torch.tensor(_filters(), device=DEV).to(torch.complex64)
    nets = [b_[2] for b_ in bands]
    sd0 = [{k: v.detach().cpu().clone() for k, v in net.state_dict().items()} for net in nets]
    bank = BandBank(nets)
    tr = BandBankTrainer(bank, _tc(), subband_filter_freq_resp=filt, band_names=CENTRES)
    assert tr._fused is not None
    sds = BandStackedDataset([b_[1] for b_ in bands])
    start, length = tr._decay_window(K)
    sds.precompute_decay_targets(4096, start, length)
    sels = [[0, 3], [1, 4]]
    seed = 31337
    step = tr.graphed(sds, B, mask_seed=seed).capture(sds.global_rows(sels))
    out = step(sds.global_rows(sels))
    torch.cuda.synchronize()
    assert int(step.mask_state.item()) == 1
    _, keep = _mask(seed, 0, length, B)
    flat = tr.optimizer.flat_grad.detach().cpu().numpy()
    views = {}
    off = 0
    for p in tr.optimizer._params:
        views[id(p)] = flat[off:off + p.numel()].reshape(tuple(p.shape))
        off += p.numel()
    N = G * NPER
    for q in range(2):
        parts_hip = {k: float(v[q]) for k, v in out.items() if k.endswith("_loss")}
        grads_hip = {"input_gains": views[id(bank.input_gains)][q].reshape(N, 1),
                     "output_gains": views[id(bank.output_gains)][q].reshape(N, 1),
                     "feedback_loop.M": views[id(bank.feedback_loop_M)][q]}
        gw, o = views[id(bank.output_scalars_w)][q], 0
        names = [n_ for n_, _ in nets[q].output_scalars.mlp.model.named_parameters()]
        for n_, prm in zip(names, bank._mlp_params[q]):
            grads_hip["output_scalars.mlp.model." + n_] = gw[o:o + prm.numel()].reshape(tuple(prm.shape))
            o += prm.numel()
        after_hip = {k: v.detach().cpu().numpy() for k, v in nets[q].state_dict().items()}
        parts, grads, after = _oracle_step(sd0[q], q, bands[q][0], bands[q][1], sels[q], filt[q], keep)
        worst = _check(f"bank[{q}]", parts_hip, grads_hip, after_hip, {k: v.numpy() for k, v in sd0[q].items()}, parts,
                       grads, after)
        for name in ("input_gains", "output_gains"):
            assert rel_err(after_hip[name], after[name].numpy()) < 1e-4, (q, name)
        print(f"bank band {q} full-size gradient deviations:", {k: f"{v:.1e}" for k, v in worst.items()})


def test_full_size_bank_distinct_decay_windows_vs_oracle():
    """Three bands whose longest decay times differ (0.6 / 1.0 / 1.5 s: EDC windows of 18 560 / 31 360 / 47 360 samples,
    reference trainer.py:56-59, run_subband_training_treble.py:286) on the timed path at full size: graph replay of the
    explicit step (slot order, pair-interleaved signals, banded EDC scans, per-band mask rows drawn on the device) -- every
    loss term and gradient of every band against the CPU oracle, whose step takes the band's own window and mask."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    from scipy.signal import firwin
    t60max = (0.6, 1.0, 1.5)
    centres = (250.0, 1000.0, 2000.0)
    delays3 = DELAYS + [[647, 719, 821, 919, 1019, 1117, 1217, 1307, 1373, 1409, 1429, 1433, 1439, 1459, 1471, 1597]]
    bands = [_band(q, delays3[q], t60max[q]) for q in range(3)]
    filt = torch.tensor(np.stack([np.fft.rfft(firwin(1025, [f / np.sqrt(2), f * np.sqrt(2)], pass_zero=False, fs=FS),
                                              n=NFFT) for f in centres]), device=DEV).to(torch.complex64)
    nets = [b_[2] for b_ in bands]
    sd0 = [{k: v.detach().cpu().clone() for k, v in net.state_dict().items()} for net in nets]
    bank = BandBank(nets)
    tr = BandBankTrainer(bank, _tc(), subband_filter_freq_resp=filt, band_names=centres)
    assert tr._fused is not None
    start, length = tr._decay_window(K)
    lens = tr._band_windows(K)
    assert lens == [int(t * FS) - start for t in t60max] and length == lens[-1]
    sds = BandStackedDataset([b_[1] for b_ in bands])
    sels = [[0, 3], [1, 4], [2, 5]]
    seed = 8128
    step = tr.graphed(sds, B, mask_seed=seed).capture(sds.global_rows(sels))
    out = step(sds.global_rows(sels))
    torch.cuda.synchronize()
    bits = philox_mask(seed, 0, length, 1.0)[0] > 0
    flat = tr.optimizer.flat_grad.detach().cpu().numpy()
    views, off = {}, 0
    for p in tr.optimizer._params:
        views[id(p)] = flat[off:off + p.numel()].reshape(tuple(p.shape))
        off += p.numel()
    N = G * NPER
    for q in range(3):
        keep = torch.argwhere(torch.tensor(bits[:lens[q]]))
        parts_hip = {k: float(v[q]) for k, v in out.items() if k.endswith("_loss")}
        grads_hip = {"input_gains": views[id(bank.input_gains)][q].reshape(N, 1),
                     "output_gains": views[id(bank.output_gains)][q].reshape(N, 1),
                     "feedback_loop.M": views[id(bank.feedback_loop_M)][q]}
        gw, o = views[id(bank.output_scalars_w)][q], 0
        names = [n_ for n_, _ in nets[q].output_scalars.mlp.model.named_parameters()]
        for n_, prm in zip(names, bank._mlp_params[q]):
            grads_hip["output_scalars.mlp.model." + n_] = gw[o:o + prm.numel()].reshape(tuple(prm.shape))
            o += prm.numel()
        after_hip = {k: v.detach().cpu().numpy() for k, v in nets[q].state_dict().items()}
        parts, grads, after = _oracle_step(sd0[q], q, bands[q][0], bands[q][1], sels[q], filt[q], keep, delays3[q])
        worst = _check(f"windows[{t60max[q]} s]", parts_hip, grads_hip, after_hip,
                       {k: v.numpy() for k, v in sd0[q].items()}, parts, grads, after)
        print(f"distinct windows, band {q} (T60max {t60max[q]} s, window {lens[q]}): gradient deviations",
              {k.replace("output_scalars.mlp.model.", "mlp."): f"{v:.1e}" for k, v in worst.items()})


@pytest.mark.parametrize("nper", [4, 8, "recipe"])
def test_full_size_bench_shape_vs_oracle(nper, monkeypatch):
    """EXACTLY what bench.py times: 7 octave bands x 32 receivers per step = 224-item launches, the 5 x 16 gain network
    on 20 Fourier features, bench.py's own workload builder, one replay of the captured explicit bank step (slot order,
    pair-interleaved signals, XCD item maps, device-drawn mask) -- every loss term of every band against the CPU oracle
    to 1e-4, every gradient against the oracle's.
    nper = 4: BASELINE.json's headline configuration, N = 16 (4 x 4), block transfer functions of csrc/blocktf.hip;
    nper = 8: ``bench.py --lines-per-group 8``, BASELINE.json configs[4], N = 32 (4 x 8): the polynomial form on the matrix
    cores (csrc/blocktf8.hip: k_tf8_pass<*>, k_tf8_rec_grads) -- the step that configuration's number is measured on.
    Gradients of the gains and of the gain network to 2e-4 of their largest entry, dL/dM to the bound DESIGN.md section 2
    derives (it is the SKEW part of the matrix-exponential adjoint of dL/d(Q Q), whose largest entry is two orders of
    magnitude above dL/dM's).
    nper = "recipe": ``bench.py --recipe reference`` -- the sub-band driver's OWN configuration
    (run_subband_training_treble.py:61-73, :105-154, :392): eight bands 63 Hz ... 8 kHz, N = 12 = 3 groups x 4 lines (every
    second band starts on an odd signal of the pair-interleaved stores), every band its own gain network (1 x 8, 1 x 16,
    5 x 16, 3 x 128) in ONE bank: k_mlp_bands_fwd / _bwd, the slot form of k_edc_lin_one."""
    import bench
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    recipe = nper == "recipe"
    G = 3 if recipe else 4                        # (shadows the module's G = 4 below)
    if recipe:
        nper = 4
        monkeypatch.setattr(bench, "RECIPE", "reference")
        monkeypatch.setattr(bench, "G", 3)
        monkeypatch.setattr(bench, "BAND_CENTRES", bench.REFERENCE_RECIPE_CENTRES)
    monkeypatch.setattr(bench, "NPER", nper)
    dev = torch.device("cuda", 0)
    R7, B7, nfeat = 40, bench.BATCH, 20
    centres = bench.BAND_CENTRES
    assert len(centres) == (8 if recipe else 7) and B7 == 32
    nets, datas, filts, rooms, delays_l = [], [], [], [], []
    for q, f in enumerate(centres):
        room, data, net, _, _, filt, delays = bench.build_workload(dev, 1234 + q, R7, centre_hz=f, room_seed=q,
                                                                   make_trainer=False)
        nets.append(net), datas.append(data), filts.append(filt), rooms.append(room), delays_l.append(delays)
    sd0 = [{k: v.detach().cpu().clone() for k, v in net.state_dict().items()} for net in nets]
    bank = BandBank(nets)
    assert bank.num_delay_lines_per_group == nper and bank.num_delay_lines == G * nper and bank.mixed_networks == recipe
    tr = BandBankTrainer(bank, bench.trainer_config(500.0, 20, train_dir="/tmp/gfdn_full/t7"),
                         subband_filter_freq_resp=torch.stack(filts), band_names=[int(f) for f in centres])
    assert tr._fused is not None                 # the explicit step, not the autograd fallback
    sds = BandStackedDataset(datas)
    start, length = tr._decay_window(K)
    sds.precompute_decay_targets(4096, start, length)
    rng = np.random.RandomState(5)
    sels = [rng.permutation(R7)[:B7].tolist() for _ in centres]
    seed = 271828
    step = tr.graphed(sds, B7, mask_seed=seed).capture(sds.global_rows(sels))
    out = step(sds.global_rows(sels))
    torch.cuda.synchronize()
    _, keep = _mask(seed, 0, length, B7)
    flat = tr.optimizer.flat_grad.detach().cpu().numpy()
    views, off = {}, 0
    for p in tr.optimizer._params:
        views[id(p)] = flat[off:off + p.numel()].reshape(tuple(p.shape))
        off += p.numel()
    N = G * nper
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))          # (the torch CPU path anti-scales beyond this)
    worst_all = {}
    # the seven oracle steps side by side in worker processes (tests/oracle_jobs.py)
    from tests.oracle_jobs import run_grid_steps
    jobs = [_oracle_job(sd0[q], q, rooms[q], datas[q], sels[q], filts[q], keep, delays_l[q], nfeat)
            for q in range(len(centres))]
    for j in jobs:
        j["G"] = G
    oracle = run_grid_steps(jobs, workers=4)
    try:
        for q in range(len(centres)):
            parts_hip = {k: float(v[q]) for k, v in out.items() if k.endswith("_loss")}
            grads_hip = {"input_gains": views[id(bank.input_gains)][q].reshape(N, 1),
                         "output_gains": views[id(bank.output_gains)][q].reshape(N, 1),
                         "feedback_loop.M": views[id(bank.feedback_loop_M)][q]}
            wflat = views[id(bank.output_scalars_w)].reshape(-1)
            gw, o = wflat[bank._w_off[q]:bank._w_off[q + 1]], 0
            names = [n_ for n_, _ in nets[q].output_scalars.mlp.model.named_parameters()]
            for n_, prm in zip(names, bank._mlp_params[q]):
                grads_hip["output_scalars.mlp.model." + n_] = gw[o:o + prm.numel()].reshape(tuple(prm.shape))
                o += prm.numel()
            after_hip = {k: v.detach().cpu().numpy() for k, v in nets[q].state_dict().items()}
            parts, grads, after = oracle[q]
            worst = _check(f"bench[N={N}, {int(centres[q])} Hz]", parts_hip, grads_hip, after_hip,
                           {k: v.numpy() for k, v in sd0[q].items()}, parts, grads, after, grad_tol=2e-4,
                           grad_tol_M=GRAD_TOL_M[nper])     # (the recipe's 4-line blocks: the N = 16 bound)
            for name in ("input_gains", "output_gains"):
                assert rel_err(after_hip[name], after[name].numpy()) < 1e-4, (q, name)
            for k, v in worst.items():
                worst_all[k] = max(worst_all.get(k, 0.0), v)
    finally:
        torch.set_num_threads(threads)
    print(f"bench shape (7 x 32, N = {N}), worst gradient deviation over the bands:",
          {k.replace("output_scalars.mlp.model.", "mlp."): f"{v:.1e}" for k, v in worst_all.items()})


def test_composed_spectra_step_equals_stored_signal_step_full_size():
    """K = 65 537, two bands: the explicit step with the EDR loss on composed short-time spectra and the EDC scans on samples
    formed on the fly (csrc/edrlin.hip: no per-receiver STFT, the receivers' signals never stored) against the same step
    with the receivers' signals stored and transformed per receiver (FusedBankStep.spectral_edr = False): the maps are
    linear and commute -- losses to 2e-6, gradients to rounding."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    res = {}
    for spec in (True, False):
        bands = [_band(q) for q in range(2)]
        filt = torch.tensor(_filters(), device=DEV).to(torch.complex64)
        bank = BandBank([b_[2] for b_ in bands])
        tr = BandBankTrainer(bank, _tc(), subband_filter_freq_resp=filt, band_names=CENTRES)
        tr._fused.spectral_edr = spec
        sds = BandStackedDataset([b_[1] for b_ in bands])
        step = tr.graphed(sds, B, mask_seed=99).capture(sds.global_rows([[0, 3], [1, 4]]))
        out = step(sds.global_rows([[2, 5], [0, 3]]))
        torch.cuda.synchronize()
        res[spec] = ({k: v.detach().cpu().numpy().copy() for k, v in out.items()},
                     tr.optimizer.flat_grad.detach().cpu().numpy().copy(),
                     [(p.numel()) for p in tr.optimizer._params])
    for k, v in res[False][0].items():
        assert np.allclose(res[True][0][k], v, rtol=2e-6, atol=0), (k, res[True][0][k], v)
    off = 0
    for i, cnt in enumerate(res[False][2]):
        a, b = res[True][1][off:off + cnt], res[False][1][off:off + cnt]
        assert np.abs(a - b).max() <= (5e-4 if i == 2 else 3e-5) * np.abs(b).max(), (i, np.abs(a - b).max(), np.abs(b).max())
        off += cnt


@pytest.mark.parametrize("per_band", [2, 4])
def test_round5_step_equals_round4_step_full_size(per_band):
    """K = 65 537, two bands, one replayed step: the step with the one-launch EDC term (csrc/edcone.hip), the normalisation
    scale joining behind the transform and the fused tail / head (k_tf_tail) against the same step with the three switched
    off (round 4's launch sequence): losses to 5e-6, gradients to rounding (dL/dM: DESIGN.md section 2).  per_band = 4: the
    wave-per-receiver gain network, with which the scale sits INSIDE the receiver gains (FusedBankStep.scale_in_gains)."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    res = {}
    sel0, sel1 = ([[0, 3], [1, 4]], [[2, 5], [0, 3]]) if per_band == 2 else ([[0, 3, 1, 5], [1, 4, 2, 0]], [[2, 5, 0, 4], [0, 3, 5, 1]])
    for new in (True, False):
        bands = [_band(q) for q in range(2)]
        filt = torch.tensor(_filters(), device=DEV).to(torch.complex64)
        bank = BandBank([b_[2] for b_ in bands])
        tr = BandBankTrainer(bank, _tc(), subband_filter_freq_resp=filt, band_names=CENTRES)
        f = tr._fused
        f.edc_one_launch = f.scale_late = f.fused_tail = new
        sds = BandStackedDataset([b_[1] for b_ in bands])
        step = tr.graphed(sds, per_band, mask_seed=99).capture(sds.global_rows(sel0))
        out = step(sds.global_rows(sel1))
        torch.cuda.synchronize()
        res[new] = ({k: v.detach().cpu().numpy().copy() for k, v in out.items()},
                    tr.optimizer.flat_grad.detach().cpu().numpy().copy(), [p.numel() for p in tr.optimizer._params])
    for k, v in res[False][0].items():
        assert np.allclose(res[True][0][k], v, rtol=5e-6, atol=0), (k, res[True][0][k], v)
    off = 0
    for i, cnt in enumerate(res[False][2]):
        a, b = res[True][1][off:off + cnt], res[False][1][off:off + cnt]
        assert np.abs(a - b).max() <= (5e-4 if i == 2 else 3e-5) * np.abs(b).max(), (i, np.abs(a - b).max(), np.abs(b).max())
        off += cnt


@pytest.mark.parametrize("per_band", [2, 4])
def test_transform_passes_equal_matrix_core_passes_full_size_n32(per_band):
    """K = 65 537, two bands of 4 x 8 lines, one replayed step: the polynomial passes of the 8-line blocks as real transforms
    of their coefficient sequences (csrc/polyfft.hip) against the same step on the matrix-core passes (csrc/blocktf8.hip):
    losses to 2e-5, gradients to the rounding of 1 / Q next to the loop's poles."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    res = {}
    # (per_band = 4: the wave-per-receiver gain network -- the scale inside the receiver gains on the transform path)
    sel0, sel1 = ([[0, 3], [1, 4]], [[2, 5], [0, 3]]) if per_band == 2 else ([[0, 3, 1, 5], [1, 4, 2, 0]], [[2, 5, 0, 4], [0, 3, 5, 1]])
    for new in (True, False):
        bands = [_band(q, delays=[d + 2 * q for d in DELAYS32]) for q in range(2)]
        filt = torch.tensor(_filters(), device=DEV).to(torch.complex64)
        bank = BandBank([b_[2] for b_ in bands])
        assert bank.num_delay_lines_per_group == 8
        tr = BandBankTrainer(bank, _tc(), subband_filter_freq_resp=filt, band_names=CENTRES)
        f = tr._fused
        f.transform_polys = new
        sds = BandStackedDataset([b_[1] for b_ in bands])
        step = tr.graphed(sds, per_band, mask_seed=99).capture(sds.global_rows(sel0))
        out = step(sds.global_rows(sel1))
        torch.cuda.synchronize()
        res[new] = ({k: v.detach().cpu().numpy().copy() for k, v in out.items()},
                    tr.optimizer.flat_grad.detach().cpu().numpy().copy(), [p.numel() for p in tr.optimizer._params])
    for k, v in res[False][0].items():
        assert np.allclose(res[True][0][k], v, rtol=2e-5, atol=0), (k, res[True][0][k], v)
    off = 0
    for i, cnt in enumerate(res[False][2]):
        a, b = res[True][1][off:off + cnt], res[False][1][off:off + cnt]
        assert np.abs(a - b).max() <= (1e-3 if i == 2 else 1e-4) * np.abs(b).max(), (i, np.abs(a - b).max(), np.abs(b).max())
        off += cnt


@pytest.mark.parametrize("nper", [4, 8])
def test_fused_tail_steps_equal_separate_launches_full_size(nper):
    """Three replayed steps with a host-launched validation step between the second and the third, the step's last launch
    leaving the next step's records (k_tf_tail / k_tf8_tail, no records launch at the head) against the same steps with
    parameter gradients, Adam and records as separate launches: the same arithmetic -- every loss of every step, the
    parameters and both Adam moments after the third step bit for bit -- and the kept records equal a fresh evaluation at the
    end.  nper = 8: blocks of 8 lines (Q, Q Q and the snapshot of the output gains are what is kept)."""
    from diffgfdn_amd import hip_ops as ops
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    res = {}
    sched = [[[2, 5], [0, 3]], [[1, 4], [2, 5]], [[0, 2], [3, 4]]]
    for fusedtail in (True, False):
        bands = [_band(q, delays=None if nper == 4 else [d + 2 * q for d in DELAYS32]) for q in range(2)]
        filt = torch.tensor(_filters(), device=DEV).to(torch.complex64)
        bank = BandBank([b_[2] for b_ in bands])
        tr = BandBankTrainer(bank, _tc(), subband_filter_freq_resp=filt, band_names=CENTRES)
        f = tr._fused
        f.fused_tail = fusedtail
        sds = BandStackedDataset([b_[1] for b_ in bands])
        step = tr.graphed(sds, B, mask_seed=99).capture(sds.global_rows([[0, 3], [1, 4]]))
        outs = []
        for i, sel in enumerate(sched):
            if i == 2:
                torch.manual_seed(5)                  # (the validation step draws its EDC mask on the host)
                _, vl = tr.valid_step(sds.collate(sds.global_rows([[1, 3], [0, 5]])))
                outs.append({k: v.detach().cpu().numpy().copy() for k, v in vl.items()})
            out = step(sds.global_rows(sel))
            torch.cuda.synchronize()
            outs.append({k: v.detach().cpu().numpy().copy() for k, v in out.items()})
        opt = tr.optimizer
        assert float(opt.step_count) == 3.0 and float(opt.step_count2) == 3.0
        if fusedtail:
            assert f.records_ok()
            if nper == 4:
                fresh = ops.tf_ortho_coefs(bank._blocks().detach(), bank.inv_gamma, bank.input_gains.data.view(-1),
                                           bank.output_gains.data.view(-1))
            else:
                fresh = (*ops.ortho_fwd(bank._blocks().detach(), True, True), bank.output_gains.data.view(-1))
            for got, want in zip(f._records(), fresh):
                assert torch.equal(got, want)
        res[fusedtail] = (outs, opt.flat_param.detach().cpu().numpy().copy(), opt.exp_avg.detach().cpu().numpy().copy(),
                          opt.exp_avg_sq.detach().cpu().numpy().copy())
    for i, (a, b) in enumerate(zip(res[True][0], res[False][0])):
        for k, v in b.items():
            assert np.array_equal(a[k], v), (i, k, a[k], v)
    for j in (1, 2, 3):
        assert np.array_equal(res[True][j], res[False][j]), j


def test_time_domain_output_stage_equals_folded_output_stage_full_size(float32_direct_store):
    """K = 65 537, two bands: the explicit step with the output stage in the time domain (4 pair-transformed group
    responses per band + the dataset's transformed direct paths, csrc/linear.hip) against the step that forms H per
    receiver inside the first pass of one transform per receiver pair (round 2/3's timed path): losses to 2e-6, gradients
    to 3e-5 of their largest entry."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    res = {}
    for lin in (True, False):
        bands = [_band(q) for q in range(2)]
        filt = torch.tensor(_filters(), device=DEV).to(torch.complex64)
        bank = BandBank([b_[2] for b_ in bands])
        tr = BandBankTrainer(bank, _tc(), subband_filter_freq_resp=filt, band_names=CENTRES)
        tr._fused.linear_transforms = lin
        sds = BandStackedDataset([b_[1] for b_ in bands])
        step = tr.graphed(sds, B, mask_seed=99).capture(sds.global_rows([[0, 3], [1, 4]]))
        out = step(sds.global_rows([[2, 5], [0, 3]]))
        torch.cuda.synchronize()
        res[lin] = ({k: v.detach().cpu().numpy().copy() for k, v in out.items()},
                    tr.optimizer.flat_grad.detach().cpu().numpy().copy(),
                    [(p.numel()) for p in tr.optimizer._params])
    for k, v in res[False][0].items():
        assert np.allclose(res[True][0][k], v, rtol=2e-6, atol=0), (k, res[True][0][k], v)
    # (flat order: output gains, input gains, M, gain network.  dL/dM is the skew part of the matrix-exponential adjoint of
    # dL/d(Q Q), two orders of magnitude below it: two float32 evaluations of the same gradient differ there by what each
    # differs from the oracle)
    off = 0
    for i, cnt in enumerate(res[False][2]):
        a, b = res[True][1][off:off + cnt], res[False][1][off:off + cnt]
        assert np.abs(a - b).max() <= (5e-4 if i == 2 else 3e-5) * np.abs(b).max(), (i, np.abs(a - b).max(), np.abs(b).max())
        off += cnt


def test_directional_full_size_forward_backward_vs_oracle():
    """BASELINE.json configs[3] at its own size (K = 65 537, irfft n = 131 072; 3 groups x 9 SH channels, 12 directions,
    2 receivers): SH-domain response, directional responses, directional EDC loss and EVERY parameter gradient of the
    module against the oracle's restatement of model.py:1043-1094, trainer.py:853-865, losses.py:333-371 under autograd
    (complex128 resolvent by torch.linalg.inv).  Runs the kernels that exist only here: 9-lane packed elimination,
    the power-of-two transform at 131 072 samples, the EDC scan against the common-slope model."""
    from diffgfdn_amd.config import CouplingMatrixType, DiffGFDNConfig, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.losses import directional_edc_loss
    from diffgfdn_amd.model import DiffDirectionalFDNVarReceiverPos
    Gd, order, J, Bd = 3, 2, 12, 2
    L = (order + 1) ** 2
    rng = np.random.RandomState(11)
    torch.manual_seed(77)
    delays = DiffGFDNConfig(num_groups=Gd, num_delay_lines=Gd * L, sample_rate=FS, seed=4711).delay_length_samps
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    A = rng.randn(J, L).astype(np.float32)
    T60 = np.linspace(0.5, 1.2, Gd)
    net = DiffDirectionalFDNVarReceiverPos(FS, Gd, delays, DEV, fl, of, ambi_order=order,
                                           common_decay_times=T60[None, :], use_colorless_loss=False,
                                           analysis_matrix=A).to(DEV)
    z = torch.exp(1j * np.pi * torch.arange(K, dtype=torch.float64) / (K - 1))
    pos = torch.tensor(rng.uniform(0, 1, (Bd, 3)))
    amps = torch.tensor(rng.uniform(0.1, 1.0, (Bd, J, Gd)))
    batch = {'z_values': z.to(DEV), 'listener_position': (10 * pos).to(DEV), 'norm_listener_position': pos.to(DEV),
             'source_position': torch.zeros(Bd, 3, dtype=torch.float64, device=DEV)}
    edc_len_ms, mix_ms = 1400.0, 20.0
    crit = directional_edc_loss(T60[None, :], edc_len_ms, FS, mixing_time_ms=mix_ms)
    H_sh = net(batch)
    Adev = net.sh_output_scalars.analysis_matrix
    from diffgfdn_amd.functional import SHToDirectional
    H_dir = SHToDirectional.apply(Adev, H_sh)
    loss = crit(H_dir, amps.to(DEV))
    loss.backward()

    # ---- oracle (CPU, float64 / complex128 with the reference's casts): oracle/cpu_trainer.directional_band_step
    from oracle.cpu_trainer import directional_band_step
    loss_o, H_sh_o, H_dir_o, grads_o = directional_band_step(net.state_dict(), delays, A, z, pos, amps, crit.envelopes.cpu(),
                                                             Gd, L, 4, orc.ms_to_samps(mix_ms, FS),
                                                             orc.ms_to_samps(edc_len_ms, FS))

    assert rel_err(H_sh.detach().cpu(), H_sh_o.detach()) < LOSS_TOL
    assert rel_err(H_dir.detach().cpu(), H_dir_o.detach()) < LOSS_TOL
    assert abs(loss.item() - loss_o.item()) < LOSS_TOL * abs(loss_o.item())
    for name, p_ in net.named_parameters():
        if name not in grads_o:
            continue
        ref = grads_o[name].numpy()
        err = np.abs(p_.grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30)
        assert err < GRAD_TOL, (name, err)

    # ---- the trainer's route: inverse transform of the SH-domain responses, analysis matrix applied to the time signals
    # (directional_edc_loss.forward_sh): same loss, same gradients
    net.zero_grad(set_to_none=True)
    loss2 = crit.forward_sh(net(batch), Adev, amps.to(DEV))
    loss2.backward()
    assert abs(loss2.item() - loss_o.item()) < LOSS_TOL * abs(loss_o.item())
    for name, p_ in net.named_parameters():
        if name not in grads_o:
            continue
        ref = grads_o[name].numpy()
        err = np.abs(p_.grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30)
        assert err < GRAD_TOL, ('forward_sh', name, err)

    # ---- the timed route: the output stage behind the transform too (directional_edc_loss.forward_lines: the 27 line
    # responses are transformed, the receivers' signals formed on the EDC window): same loss, same gradients
    net.zero_grad(set_to_none=True)
    assert crit.lines_supported(K, Gd, L, J, Gd)
    Y, c, w = net.output_stage_inputs(batch)
    loss3 = crit.forward_lines(Y, c, w, Gd, L, None, Adev, amps.to(DEV))
    loss3.backward()
    assert abs(loss3.item() - loss_o.item()) < LOSS_TOL * abs(loss_o.item())
    for name, p_ in net.named_parameters():
        if name not in grads_o:
            continue
        ref = grads_o[name].numpy()
        err = np.abs(p_.grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30)
        assert err < GRAD_TOL, ('forward_lines', name, err)


def test_directional_full_size_trainer_step_with_colorless_terms_vs_oracle():
    """The directional trainer's own step at BASELINE.json configs[3]'s size WITH the colorless terms (reference
    trainer.py:690-921, :298-313), as bench.py times it: line responses -> 27 line transforms -> directional EDC loss on
    the window (csrc/dirlin.hip, csrc/edcmix.hip) on the main stream, the raw sub-FDN branch with the spectral and sparsity
    terms on the side stream; every term and every parameter gradient against the oracle's autograd."""
    from diffgfdn_amd.config import (CouplingMatrixType, DiffGFDNConfig, FeedbackLoopConfig, OutputFilterConfig,
                                     TrainerConfig)
    from diffgfdn_amd.model import DiffDirectionalFDNVarReceiverPos
    from diffgfdn_amd.trainer import DirectionalFDNVarReceiverPosTrainer
    Gd, order, J, Bd = 3, 2, 12, 2
    L = (order + 1) ** 2
    rng = np.random.RandomState(13)
    torch.manual_seed(78)
    delays = DiffGFDNConfig(num_groups=Gd, num_delay_lines=Gd * L, sample_rate=FS, seed=4712).delay_length_samps
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    A = rng.randn(J, L).astype(np.float32)
    T60 = np.linspace(0.5, 1.2, Gd)
    net = DiffDirectionalFDNVarReceiverPos(FS, Gd, delays, DEV, fl, of, ambi_order=order,
                                           common_decay_times=T60[None, :], use_colorless_loss=True,
                                           analysis_matrix=A).to(DEV)
    tc = TrainerConfig(use_colorless_loss=True, use_asym_spectral_loss=True, edc_loss_weight=10.0,
                       spectral_loss_weight=1.5, sparsity_loss_weight=2.0, use_edc_mask=False, lr=1e-3, io_lr=1e-2,
                       device='cuda', train_dir='/tmp/gfdn_t/dirc', ir_dir='/tmp/gfdn_a/dirc')
    tr = DirectionalFDNVarReceiverPosTrainer(net, tc)
    z = torch.exp(1j * np.pi * torch.arange(K, dtype=torch.float64) / (K - 1))
    pos = torch.tensor(rng.uniform(0, 1, (Bd, 3)))
    amps = torch.tensor(rng.uniform(0.1, 1.0, (Bd, J, Gd)))
    batch = {'z_values': z.to(DEV), 'listener_position': (10 * pos).to(DEV), 'norm_listener_position': pos.to(DEV),
             'source_position': torch.zeros(Bd, 3, dtype=torch.float64, device=DEV),
             'target_common_slope_amps': amps.to(DEV)}
    assert tr._use_lines(batch) and tr._side_stream() is not None
    crit = tr.criterion[0]
    net.zero_grad(set_to_none=True)
    losses = tr._step_losses(batch)
    total = losses.pop('_total')
    total.backward()
    torch.cuda.synchronize()

    from oracle.cpu_trainer import directional_band_step
    tot_o, _, _, grads_o, terms_o = directional_band_step(
        net.state_dict(), delays, A, z, pos, amps, crit.envelopes.cpu(), Gd, L, 4, crit.mixing_time_samps,
        crit.edc_len_samps, edc_weight=10.0,
        colorless={'spectral_weight': 1.5, 'sparsity_weight': 2.0, 'use_asym': True})
    for k, v in terms_o.items():
        assert abs(float(losses[k]) - float(v)) < LOSS_TOL * abs(float(v)), (k, float(losses[k]), float(v))
    assert abs(float(total.detach()) - float(tot_o)) < LOSS_TOL * abs(float(tot_o))
    for name, p_ in net.named_parameters():
        if name not in grads_o:
            continue
        ref = grads_o[name].numpy()
        err = np.abs(p_.grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30)
        assert err < GRAD_TOL, (name, err)
