"""Module-level parity on the MI355X: the diffgfdn_amd nn.Modules / losses / trainer step through
the C-ABI against the golden vectors generated from the reference (tests/golden/*.npz) and
against the float64 oracle.  Tolerance: 1e-4 relative (north star), tighter where it holds."""
import numpy as np
import pytest
import torch

from oracle import gfdn_oracle as orc
from tests.helpers import batch_from, load, rel_err
from tests.margins import within

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _to_dev(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


def _state(fx, prefix="sd_"):
    return {k[len(prefix):]: torch.tensor(v) for k, v in fx.items() if k.startswith(prefix)}


def _grid_model(fx, zero=None, layers=2, neurons=16, nff=4):
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    zero = bool(fx["zero_coupling"]) if zero is None else zero
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=zero)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=layers, num_neurons_per_layer=neurons,
                            num_fourier_features=nff)
    net = DiffGFDNVarReceiverPos(float(fx["fs"]), int(fx["G"]), fx["delays"].tolist(), DEV, fl, of,
                                 use_absorption_filters=False, common_decay_times=fx["T60"][None, :],
                                 use_colorless_loss=True)
    missing, unexpected = net.load_state_dict(_state(fx), strict=True)
    return net.to(DEV)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["zc", "cp", "g1"])
def test_f1_feedback_loop(tag):
    from diffgfdn_amd.config import CouplingMatrixType
    from diffgfdn_amd.feedback_loop import FeedbackLoop
    fx = load("f1_feedback_loop.npz")
    M = fx[f"{tag}_M"]
    G, n, _ = M.shape
    loop = FeedbackLoop(float(fx["fs"]), G, n, torch.tensor(fx[f"{tag}_delays"], dtype=torch.float32),
                        False, coupling_matrix_type=CouplingMatrixType.SCALAR,
                        use_zero_coupling=(tag != "cp"), gains=torch.tensor(fx[f"{tag}_gamma"]))
    with torch.no_grad():
        loop.M.copy_(torch.tensor(M))
        loop.alpha.copy_(torch.tensor(fx[f"{tag}_alpha"]))
    loop = loop.to(DEV)
    z = torch.tensor(fx["z"]).to(DEV)
    P = loop(z)
    assert rel_err(loop.get_coupled_feedback_matrix().real.detach().cpu(), fx[f"{tag}_A"]) < 1e-5
    assert rel_err(P.detach().cpu(), fx[f"{tag}_P"]) < TOL
    (P.abs() ** 2).sum().backward()
    within(rel_err(loop.M.grad.cpu(), fx[f"{tag}_grad_M"]), 2e-4, "L65")
    if tag == "cp":
        within(rel_err(loop.alpha.grad.cpu(), fx[f"{tag}_grad_alpha"]), 2e-5, "L67")


@pytest.mark.parametrize("name", ["f234_n12_k257.npz", "f234_n16_k4097_cp.npz", "f234_n32_k1025.npz"])
def test_f2_model_forward(name):
    fx = load(name)
    net = _grid_model(fx)
    batch = _to_dev(batch_from(fx))
    H, (Hout, Hpd) = net(batch)
    assert rel_err(net.output_scalars.gains.detach().cpu(), fx["receiver_gains"]) < 1e-5
    assert rel_err(H.detach().cpu(), fx["H"]) < TOL
    assert rel_err(Hout.detach().cpu(), fx["Hout"]) < TOL
    n, G = int(fx["nper"]), int(fx["G"])
    assert Hpd.shape == (n * G, H.shape[-1], G)
    for g in range(G):
        assert rel_err(Hpd[g * n:(g + 1) * n, :, g].detach().cpu(), fx["Hout_per_del_nz"][g]) < TOL
    # off-group entries are exactly zero, like the reference's torch.zeros buffer
    assert float(Hpd[0:n, :, 1].abs().max()) == 0.0


@pytest.mark.parametrize("name", ["f234_n12_k257.npz", "f234_n16_k4097_cp.npz", "f234_n32_k1025.npz"])
def test_f3_losses(name):
    from diffgfdn_amd.colorless_losses import amse_loss, mse_loss, sparsity_loss
    from diffgfdn_amd.losses import edc_loss, edr_loss
    fx = load(name)
    fs = float(fx["fs"])
    tgt = torch.tensor(fx["batch_target_rir_response"]).to(DEV)
    H = torch.tensor(fx["H"]).to(DEV).requires_grad_(True)
    l = edr_loss(fs, win_size=int(fx["win"]), hop_size=int(fx["hop"]))(tgt, H)
    g, = torch.autograd.grad(l, H)
    assert abs(l.item() - float(fx["loss_edr"])) < TOL * abs(float(fx["loss_edr"]))
    within(np.abs(g.cpu().numpy() - fx["grad_edr_H"]).sum() / np.abs(fx["grad_edr_H"]).sum(), 1e-4, "L98")
    l = edc_loss(float(np.max(fx["T60"])) * 1e3, fs, use_mask=False)(tgt, H)
    g, = torch.autograd.grad(l, H)
    assert abs(l.item() - float(fx["loss_edc"])) < TOL * abs(float(fx["loss_edc"]))
    within(np.abs(g.cpu().numpy() - fx["grad_edc_H"]).sum() / np.abs(fx["grad_edc_H"]).sum(), 5e-5, "L102")
    Hout = torch.tensor(fx["Hout"]).to(DEV).requires_grad_(True)
    for nm, crit in (("mse", mse_loss()), ("amse", amse_loss())):
        for k in range(int(fx["G"])):
            l = crit(Hout[..., k], torch.ones_like(Hout[..., k]))
            g, = torch.autograd.grad(l, Hout)
            assert abs(l.item() - fx[f"loss_{nm}"][k]) < TOL * abs(fx[f"loss_{nm}"][k])
            assert rel_err(g[..., k].cpu(), fx[f"grad_{nm}_Hout"][k]) < TOL
    from diffgfdn_amd.feedback_loop import MatrixExponential, Skew
    M = torch.tensor(fx["sd_feedback_loop.M"]).to(DEV)
    for k in range(int(fx["G"])):
        Q = MatrixExponential()(Skew()(M[k]))
        assert abs(sparsity_loss()(Q).item() - fx["loss_sparsity"][k]) < 1e-5


@pytest.mark.parametrize("name,asym", [("f234_n12_k257.npz", True), ("f234_n16_k4097_cp.npz", False),
                                       ("f234_n32_k1025.npz", True)])
def test_f4_train_step(name, asym):
    """normalize + fused step + Adam against the reference trainer (values, grads, new state)."""
    from diffgfdn_amd.config import TrainerConfig
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    fx = load(name)
    net = _grid_model(fx)
    batch = _to_dev(batch_from(fx))
    tc = TrainerConfig(batch_size=4, num_freq_bins=int(fx["nfft"]), max_epochs=1, lr=1e-3, io_lr=1e-2,
                       coupling_angle_lr=1e-2, use_colorless_loss=True, use_asym_spectral_loss=asym,
                       edc_loss_weight=10.0, edr_loss_weight=1.0, spectral_loss_weight=1.0,
                       sparsity_loss_weight=2.0, use_edc_mask=False, train_dir="/tmp/gfdn_t",
                       ir_dir="/tmp/gfdn_a", device="cuda")
    tr = VarReceiverPosTrainer(net, tc, stft_win=int(fx["win"]))
    tr.normalize(batch)
    assert rel_err(net.input_gains.detach().cpu(), fx["sdn_input_gains"]) < TOL
    assert rel_err(net.output_gains.detach().cpu(), fx["sdn_output_gains"]) < TOL
    # fused path
    losses = tr._step_losses(batch)
    total = losses.pop("_total")
    for k, v in losses.items():
        ref = float(fx["step_" + k])
        assert abs(float(v) - ref) < TOL * abs(ref) + 1e-7, (k, float(v), ref)
    assert abs(total.item() - float(fx["step_total"])) < TOL * abs(float(fx["step_total"]))
    total.backward()
    for name_, prm in net.named_parameters():
        ref = fx["grad_" + name_]
        got = prm.grad.cpu().numpy()
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)
        within(err, 2e-4, ("L147",) + tuple((name_, err)))
    tr.optimizer.step()
    for name_, prm in net.named_parameters():
        assert rel_err(prm.detach().cpu(), fx["sda_" + name_]) < 1e-4, name_
    # drop-in path (separate loss modules, reference calculate_losses) agrees with the fused one
    net2 = _grid_model(fx)
    tr2 = VarReceiverPosTrainer(net2, tc, stft_win=int(fx["win"]))
    tr2.normalize(batch)
    H, Hs = net2(batch)
    d = tr2.calculate_losses(batch, H, Hs)
    for k, v in d.items():
        ref = float(fx["step_" + k])
        assert abs(float(v) - ref) < TOL * abs(ref) + 1e-7, (k, float(v), ref)


def test_f14_learnable_decay_times():
    """Learnable common decay times (feedback_loop.py:205-232): H, the decay losses and every gradient incl.
    dL/dT60 of the MODULE against the reference's first forward / backward (fixture F14)."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.losses import edc_loss, edr_loss
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    fx = load("f14_learnable_decay_times.npz")
    fs = float(fx["fs"])
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    net = DiffGFDNVarReceiverPos(fs, int(fx["G"]), fx["delays"].tolist(), DEV, fl, of, use_absorption_filters=False,
                                 learn_common_decay_times=True, common_decay_times=fx["T60"][None, :],
                                 use_colorless_loss=False)
    net.load_state_dict(_state(fx), strict=True)
    net = net.to(DEV)
    batch = _to_dev(batch_from(fx))
    H = net(batch)
    assert rel_err(H.detach().cpu(), fx["H"]) < TOL
    tgt = batch["target_rir_response"]
    l_edr = edr_loss(fs, win_size=int(fx["win"]), hop_size=int(fx["hop"]))(tgt, H)
    l_edc = edc_loss(float(np.max(fx["T60"])) * 1e3, fs, use_mask=False)(tgt, H)
    assert abs(l_edr.item() - float(fx["loss_edr"])) < TOL * abs(float(fx["loss_edr"]))
    assert abs(l_edc.item() - float(fx["loss_edc"])) < TOL * abs(float(fx["loss_edc"]))
    (l_edr + 10.0 * l_edc).backward()
    for name_, prm in net.named_parameters():
        ref = fx["grad_" + name_]
        err = np.abs(prm.grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30)
        within(err, 1e-4, ("L189",) + tuple((name_, err)))
    assert net.feedback_loop.common_decay_times.grad is not None


def test_graphed_step_equals_eager_step():
    """HIP-graph replay of normalize + train_step == the eager step (same masks, same state)."""
    from diffgfdn_amd.config import TrainerConfig
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset
    from diffgfdn_amd.synthetic import synthetic_room
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    fx = load("f234_n16_k4097_cp.npz")
    room = synthetic_room(12, 4, 8000.0, 5000, seed=2)
    ds = MultiRIRDataset(DEV, RoomDataset(4, 8000.0, room["source_position"], room["receiver_position"],
                                          room["rirs"], room["common_decay_times"], nfft=8192, device=DEV))
    tc = TrainerConfig(batch_size=4, num_freq_bins=8192, lr=1e-3, io_lr=1e-2, coupling_angle_lr=1e-2,
                       use_colorless_loss=True, use_asym_spectral_loss=True, edc_loss_weight=10.0,
                       sparsity_loss_weight=2.0, use_edc_mask=True, train_dir="/tmp/gfdn_t", ir_dir="/tmp/gfdn_a",
                       device="cuda")
    results = []
    for mode in ("eager", "graph"):
        net = _grid_model(fx)
        tr = VarReceiverPosTrainer(net, tc, stft_win=512, capturable=(mode == "graph"))
        if mode == "graph":
            step = tr.graphed(ds, 4, mask_source="host").capture([0, 1, 2, 3])
        torch.manual_seed(77)
        tot = []
        for sel in ([0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11]):
            if mode == "eager":
                b = ds.collate(sel)
                tr.normalize(b)
                t, _ = tr.train_step(b)
                tot.append(float(t))
            else:
                tot.append(float(step(sel)["_total"]))
        results.append((tot, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
    (t0, s0), (t1, s1) = results
    for a, b in zip(t0, t1):
        assert abs(a - b) < 1e-5 * abs(a), (t0, t1)
    # Adam divides by sqrt(v): near-zero gradients (biases) amplify float-level differences between
    # the capturable and the default update, so the state is compared at 5e-4
    for k in s0:
        within(rel_err(s1[k], s0[k]), 1e-4, ("L230",) + tuple((k,)))


def test_graphed_step_device_mask():
    """The graph draws the EDC time mask itself (gfdn_draw_mask): every replay must equal an eager
    step that is handed the mask the Philox restatement predicts for (seed, replay number)."""
    from diffgfdn_amd.config import TrainerConfig
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset
    from diffgfdn_amd.synthetic import synthetic_room
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    from tests.helpers import philox_mask
    fx = load("f234_n16_k4097_cp.npz")
    room = synthetic_room(12, 4, 8000.0, 5000, seed=2)
    ds = MultiRIRDataset(DEV, RoomDataset(4, 8000.0, room["source_position"], room["receiver_position"],
                                          room["rirs"], room["common_decay_times"], nfft=8192, device=DEV))
    tc = TrainerConfig(batch_size=4, num_freq_bins=8192, lr=1e-3, io_lr=1e-2, coupling_angle_lr=1e-2,
                       use_colorless_loss=True, use_asym_spectral_loss=True, edc_loss_weight=10.0,
                       sparsity_loss_weight=2.0, use_edc_mask=True, train_dir="/tmp/gfdn_t", ir_dir="/tmp/gfdn_a",
                       device="cuda")
    sels = ([0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11])
    net = _grid_model(fx)
    tr = VarReceiverPosTrainer(net, tc, stft_win=512, capturable=True)
    step = tr.graphed(ds, 4, mask_seed=4242).capture(sels[0])
    assert int(step.mask_state.item()) == 0                     # the warm-up left no trace
    got, got_grad = [], []
    for sel in sels:
        got.append(float(step(sel)["_total"]))
        got_grad.append(tr.optimizer.flat_grad.detach().cpu().numpy().copy())
    assert int(step.mask_state.item()) == len(sels)
    last = step.maskw.cpu().numpy()
    want_last, _ = philox_mask(4242, len(sels) - 1, step.length, 1.0 / 4)
    assert np.array_equal(last, want_last)

    net2 = _grid_model(fx)
    tr2 = VarReceiverPosTrainer(net2, tc, stft_win=512, capturable=True)
    want = []
    for i, sel in enumerate(sels):
        b = ds.collate(sel)
        tr2.normalize(b)
        tr2.optimizer.zero_grad(set_to_none=True)
        mw = torch.tensor(philox_mask(4242, i, step.length, 1.0 / 4)[0], device=DEV)
        losses = tr2._step_losses(b, mask_prenorm=mw)
        losses["_total"].backward()
        tr2.optimizer.pack_grads()
        if i == 0:      # same state and inputs: replayed gradients equal the eager ones to the last bit
            assert np.array_equal(tr2.optimizer.flat_grad.cpu().numpy(), got_grad[0])
        tr2.optimizer.step()
        want.append(float(losses["_total"]))
    for a, b in zip(got, want):
        assert abs(a - b) < 1e-5 * abs(b), (got, want)
    for k, v in net.state_dict().items():
        within(rel_err(v.detach().cpu(), net2.state_dict()[k].detach().cpu()), 1e-6, ("L281",) + tuple((k,)))


def test_f3b_subband_mask_weights():
    from diffgfdn_amd.losses import edc_loss, edr_loss
    fx = load("f3b_subband_mask.npz")
    fs = float(fx["fs"])
    H = torch.tensor(fx["H"]).to(DEV).to(torch.complex64).requires_grad_(True)
    tgt = torch.tensor(fx["target"]).to(DEV)
    filt = torch.tensor(fx["filt"]).to(DEV).to(torch.complex64)
    Hs = H * filt
    crit = edr_loss(fs, win_size=256, hop_size=128, use_weight_fn=True)
    assert rel_err(crit.frequency_weights, fx["freq_weights"]) < 1e-12
    l = crit(tgt, Hs)
    g, = torch.autograd.grad(l, H, retain_graph=True)
    assert abs(l.item() - float(fx["loss_edr_w"])) < TOL * abs(float(fx["loss_edr_w"]))
    within(np.abs(g.cpu().numpy() - fx["grad_edr_w"]).sum() / np.abs(fx["grad_edr_w"]).sum(), 1e-3, "L297")
    crit2 = edc_loss(float(np.max(fx["T60"])) * 1e3, fs, use_mask=True)
    l = crit2(tgt, Hs, mask_index=torch.tensor(fx["edc_mask_index"]))
    g, = torch.autograd.grad(l, H)
    assert abs(l.item() - float(fx["loss_edc_masked"])) < TOL * abs(float(fx["loss_edc_masked"]))
    within(np.abs(g.cpu().numpy() - fx["grad_edc_masked"]).sum() / np.abs(fx["grad_edc_masked"]).sum(), 2e-4, "L302")
    # the random mask is drawn from the global CPU generator exactly like the reference
    torch.manual_seed(99)
    l2 = crit2(tgt, Hs)
    assert abs(l2.item() - float(fx["loss_edc_masked"])) < TOL * abs(float(fx["loss_edc_masked"]))


def test_f5_single_pos():
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.losses import edr_loss
    from diffgfdn_amd.model import DiffGFDNSinglePos
    fx = load("f5_single_pos.npz")
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=False)
    net = DiffGFDNSinglePos(float(fx["fs"]), int(fx["G"]), fx["delays"].tolist(), DEV, fl,
                            OutputFilterConfig(use_svfs=False), use_absorption_filters=False,
                            common_decay_times=fx["T60"][None, :], use_colorless_loss=True)
    net.load_state_dict(_state(fx), strict=True)
    net = net.to(DEV)
    x = {"z_values": torch.tensor(fx["z"]).to(DEV), "target_early_response": torch.tensor(fx["early"]).to(DEV)}
    H, (Hout, _) = net(x)
    assert H.shape == (len(fx["z"]),)
    assert rel_err(H.detach().cpu(), fx["H"]) < TOL
    assert rel_err(Hout.detach().cpu(), fx["Hout"]) < TOL
    Hd = torch.tensor(fx["H"]).to(DEV).requires_grad_(True)
    l = edr_loss(float(fx["fs"]), win_size=256, hop_size=128)(torch.tensor(fx["target"]).to(DEV), Hd)
    assert abs(l.item() - float(fx["loss_edr"])) < TOL * abs(float(fx["loss_edr"]))


def test_f6_directional():
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.losses import directional_edc_loss
    from diffgfdn_amd.model import DiffDirectionalFDNVarReceiverPos
    fx = load("f6_directional.npz")
    fs = float(fx["fs"])
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=1, num_neurons_per_layer=8,
                            num_fourier_features=3)
    net = DiffDirectionalFDNVarReceiverPos(fs, int(fx["G"]), fx["delays"].tolist(), DEV, fl, of,
                                           ambi_order=int(fx["order"]), common_decay_times=fx["T60"][None, :],
                                           use_colorless_loss=True, analysis_matrix=fx["analysis_matrix"])
    net.load_state_dict(_state(fx), strict=True)
    net = net.to(DEV)
    batch = _to_dev(batch_from(fx))
    H_sh, (Hout, _) = net(batch)
    assert rel_err(net.sh_output_scalars.weights.detach().cpu(), fx["sh_gains"]) < 1e-5
    assert rel_err(H_sh.detach().cpu(), fx["H_sh"]) < TOL
    assert rel_err(Hout.detach().cpu(), fx["Hout"]) < TOL
    A = net.sh_output_scalars.analysis_matrix
    H_dir = torch.einsum('jl, blk -> bjk', torch.complex(A, torch.zeros_like(A)), H_sh)
    assert rel_err(H_dir.detach().cpu(), fx["H_dir"]) < TOL
    crit = directional_edc_loss(fx["T60"][None, :], float(fx["edc_len_ms"]), fs,
                                envelopes=torch.tensor(fx["envelopes"]))
    loss = crit(H_dir, torch.tensor(fx["amps"]).to(DEV))
    assert abs(loss.item() - float(fx["loss"])) < TOL * abs(float(fx["loss"]))
    loss.backward()
    for name_, prm in net.named_parameters():
        key = "grad_" + name_
        if key in fx:
            err = np.abs(prm.grad.cpu().numpy() - fx[key]).max() / (np.abs(fx[key]).max() + 1e-30)
            within(err, 1e-4, ("L361",) + tuple((name_, err)))
    # the default envelope formula equals the fixture's stated one
    crit_default = directional_edc_loss(fx["T60"][None, :], float(fx["edc_len_ms"]), fs)
    assert rel_err(crit_default.envelopes, fx["envelopes"]) < 1e-5


@pytest.mark.parametrize("weighted,radius", [(False, None), (True, 1.0002)])
def test_edr_loss_with_erb_grouping(weighted, radius):
    """edr_loss(use_erb_grouping=True) (reference losses.py:430-495 with :545-551: the EDR of erb_filters |STFT|) against
    the oracle, value and dloss/dH; the band matrix is an input (librosa is absent), here the restated Slaney mel bank."""
    from diffgfdn_amd.losses import edr_loss, mel_filterbank, mel_frequencies
    fs, win, K, B = 16000.0, 512, 4097, 3
    g = torch.Generator().manual_seed(4)
    t = torch.arange(K, dtype=torch.float64)
    rir_t = torch.randn(B, K, generator=g, dtype=torch.float64) * torch.exp(-t / 900.0)
    rir_a = torch.randn(B, K, generator=g, dtype=torch.float64) * torch.exp(-t / 700.0)
    # spectra whose irfft(n = K) are those signals: the first (K + 1) / 2 bins of their length-K DFT, zero above
    def spec(x):
        X = torch.zeros(B, K, dtype=torch.complex128)
        X[:, :(K + 1) // 2] = torch.fft.rfft(x, n=K)
        return X
    Ht, Ha = spec(rir_t), spec(rir_a)
    erb = mel_filterbank(fs, win, 16, 63.0, 7000.0)
    crit = edr_loss(fs, win_size=win, hop_size=win // 2, use_erb_grouping=True, erb_filters=erb,
                    reduced_pole_radius=radius)
    wf = None
    if weighted:
        f = torch.tensor(mel_frequencies(16, 63.0, 7000.0))
        wf = 2.0 + (1.0 - 2.0) / (1 + torch.exp(10 ** (-2.5) * (f - 1e3)))
        crit.use_weight_fn, crit.frequency_weights = True, wf
    Hr = Ha.clone().requires_grad_()
    want = orc.edr_loss(Ht, Hr, win, win // 2, radius, wf, torch.tensor(erb))
    want.backward()
    Hd = Ha.to(torch.complex64).to(DEV).requires_grad_()
    got = crit(Ht.to(torch.complex64).to(DEV), Hd)
    got.backward()
    assert abs(got.item() - want.item()) < TOL * abs(want.item())
    gr = Hr.grad[:, :(K + 1) // 2]
    gd = Hd.grad.cpu().to(torch.complex128)[:, :(K + 1) // 2]
    within(float((gd - gr).abs().sum() / gr.abs().sum()), 3e-4, "L400")      # (L1: sign flips of the |.| at tiny differences)


@pytest.mark.parametrize("mask", [False, True])
def test_directional_graphed_step_equals_eager_step(mask):
    """DirectionalFDNVarReceiverPosTrainer.graphed: replaying the captured step == host launches (values, state) -- also
    with the random EDC time mask (reference losses.py:355-360), which the captured step draws on the device."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig, TrainerConfig
    from diffgfdn_amd.model import DiffDirectionalFDNVarReceiverPos
    from diffgfdn_amd.trainer import DirectionalFDNVarReceiverPosTrainer
    fx = load("f6_directional.npz")
    fs = float(fx["fs"])
    res = {}
    batch0 = _to_dev(batch_from(fx))
    batch0["target_common_slope_amps"] = torch.tensor(fx["amps"]).to(DEV)
    for mode in ("eager", "graph"):
        fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
        of = OutputFilterConfig(use_svfs=False, num_hidden_layers=1, num_neurons_per_layer=8, num_fourier_features=3)
        net = DiffDirectionalFDNVarReceiverPos(fs, int(fx["G"]), fx["delays"].tolist(), DEV, fl, of,
                                               ambi_order=int(fx["order"]), common_decay_times=fx["T60"][None, :],
                                               use_colorless_loss=True, analysis_matrix=fx["analysis_matrix"])
        net.load_state_dict(_state(fx), strict=True)
        net = net.to(DEV)
        tc = TrainerConfig(use_colorless_loss=True, edc_loss_weight=1.0, use_edc_mask=mask, lr=1e-3, io_lr=1e-2,
                           train_dir="/tmp/gfdn_t", ir_dir="/tmp/gfdn_a", device="cuda")
        tr = DirectionalFDNVarReceiverPosTrainer(net, tc, capturable=True)
        crit = tr.criterion[0]
        crit.edc_len_samps = int(float(fx["edc_len_ms"]) * 1e-3 * fs)
        crit.envelopes = torch.tensor(fx["envelopes"], dtype=torch.float32)
        assert crit.use_mask == mask
        vals = []
        step = tr.graphed(batch0, mask_seed=4242) if mode == "graph" else None
        if mask and mode == "eager":          # the same device generator, launched from the host
            K = batch0["z_values"].shape[-1]
            L = min(crit.edc_len_samps, 2 * (K - 1) - crit.mixing_time_samps)
            crit.device_mask = (4242, torch.zeros(1, dtype=torch.long, device=DEV), torch.zeros(L, device=DEV))
        for i in range(3):
            b = dict(batch0)
            b["target_common_slope_amps"] = batch0["target_common_slope_amps"] * (1.0 + 0.1 * i)
            total, _ = step(b) if step is not None else tr.train_step(b)
            vals.append(float(total))
        res[mode] = (vals, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    assert np.allclose(res["eager"][0], res["graph"][0], rtol=1e-5), res
    for k, v in res["eager"][1].items():
        assert rel_err(res["graph"][1][k], v) < 1e-5, k
    if mask:
        assert int(step.mask_state.item()) == 3          # (one draw per replay; the warm-up draws were undone)


@pytest.mark.parametrize("mask", [False, True])
def test_directional_bank_equals_band_steps(mask):
    """trainer.DirectionalBank: three bands' directional trainers (different parameters, targets and decay times) stepped by
    ONE graph with the bands on two lanes == every band's own host-launched train_step, bit for bit (values and state) --
    the bands are independent models, the graph only changes who launches what beside what."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig, TrainerConfig
    from diffgfdn_amd.model import DiffDirectionalFDNVarReceiverPos
    from diffgfdn_amd.trainer import DirectionalBank, DirectionalFDNVarReceiverPosTrainer
    fx = load("f6_directional.npz")
    fs = float(fx["fs"])
    batch0 = _to_dev(batch_from(fx))
    nb = 3

    def build():
        trs, batches = [], []
        for q in range(nb):
            torch.manual_seed(900 + q)
            fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
            of = OutputFilterConfig(use_svfs=False, num_hidden_layers=1, num_neurons_per_layer=8, num_fourier_features=3)
            net = DiffDirectionalFDNVarReceiverPos(fs, int(fx["G"]), fx["delays"].tolist(), DEV, fl, of,
                                                   ambi_order=int(fx["order"]),
                                                   common_decay_times=(fx["T60"] * (1.0 + 0.15 * q))[None, :],
                                                   use_colorless_loss=True, analysis_matrix=fx["analysis_matrix"])
            sd = _state(fx)
            if q:                                          # (band q > 0: the fixture's parameters, perturbed)
                g = torch.Generator().manual_seed(77 + q)
                sd = {k: (v + 0.05 * torch.randn(v.shape, generator=g).to(v.dtype) if v.is_floating_point() and
                          k in ("input_gains", "output_gains", "feedback_loop.M") else v) for k, v in sd.items()}
            net.load_state_dict(sd, strict=True)
            net = net.to(DEV)
            tc = TrainerConfig(use_colorless_loss=True, edc_loss_weight=1.0 + q, use_edc_mask=mask, lr=1e-3, io_lr=1e-2,
                               train_dir=f"/tmp/gfdn_t/db{q}", ir_dir=f"/tmp/gfdn_a/db{q}", device="cuda")
            tr = DirectionalFDNVarReceiverPosTrainer(net, tc, capturable=True)
            crit = tr.criterion[0]
            crit.edc_len_samps = int(float(fx["edc_len_ms"]) * 1e-3 * fs)
            b = dict(batch0)
            b["target_common_slope_amps"] = torch.tensor(fx["amps"]).to(DEV) * (1.0 + 0.2 * q)
            trs.append(tr), batches.append(b)
        return trs, batches

    res = {}
    for mode in ("eager", "bank"):
        trs, batches = build()
        bank = DirectionalBank(trs, batches, lanes=2, mask_seed=977).capture() if mode == "bank" else None
        if mask and mode == "eager":          # (the bank's device generators, band q seeded 977 + q, launched from the host)
            for q, tr in enumerate(trs):
                crit = tr.criterion[0]
                K_ = batches[q]["z_values"].shape[-1]
                L_ = min(crit.edc_len_samps, 2 * (K_ - 1) - crit.mixing_time_samps)
                crit.device_mask = (977 + q, torch.zeros(1, dtype=torch.long, device=DEV), torch.zeros(L_, device=DEV))
        vals = []
        for i in range(3):
            bs = [dict(b, target_common_slope_amps=b["target_common_slope_amps"] * (1.0 + 0.1 * i)) for b in batches]
            outs = bank(bs) if bank is not None else [tr.train_step(b) for tr, b in zip(trs, bs)]
            torch.cuda.synchronize()
            vals.append([float(o[0]) for o in outs])
        res[mode] = (vals, [{k: v.detach().cpu().clone() for k, v in tr.net.state_dict().items()} for tr in trs])
    assert len({round(v, 3) for v in res["eager"][0][0]}) == nb          # (the bands really differ)
    assert np.allclose(res["eager"][0], res["bank"][0], rtol=1e-6, atol=0), res
    for q in range(nb):
        for k, v in res["eager"][1][q].items():
            assert rel_err(res["bank"][1][q][k], v) < 1e-6, (q, k)


@pytest.mark.parametrize("tag", ["zc", "cp", "mixed"])
def test_f16_source_receiver_svf_model(tag):
    """DiffGFDNVarSourceReceiverPos with SVF filters from MLPs (reference model.py:347-452): the reference's state dict
    loads with strict=True; H, the sub-FDN responses and every parameter gradient of the reference's forward / backward.
    zc: zero coupling, both sides SVF -- one 22-section cascade per (position, group) inside the contraction kernel;
    cp: learnable coupling (G x G group transfer functions, responses contracted by tensor operations);
    mixed: SVF output filters, scalar input gains (the gain scales the cascade's first numerator)."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNVarSourceReceiverPos
    fx = load(f"f16_source_receiver_svf_{tag}.npz")
    fs, G = float(fx["fs"]), int(fx["G"])
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=bool(fx["zero_coupling"]))
    of = OutputFilterConfig(use_svfs=True, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4,
                            compress_pole_factor=0.98)
    inf = OutputFilterConfig(use_svfs=bool(fx["svf_in"]), num_hidden_layers=2, num_neurons_per_layer=16,
                             num_fourier_features=4, compress_pole_factor=0.98)
    net = DiffGFDNVarSourceReceiverPos(fs, G, fx["delays"].tolist(), DEV, fl, of, inf, use_absorption_filters=False,
                                       learn_common_decay_times=False, common_decay_times=fx["T60"][None, :],
                                       use_colorless_loss=True)
    net.load_state_dict(_state(fx), strict=True)
    net = net.to(DEV)
    batch = _to_dev(batch_from(fx))
    H, (Hout, _) = net(batch)
    assert rel_err(H.detach().cpu(), fx["H"]) < 1e-4
    assert rel_err(Hout.detach().cpu(), fx["Hout"]) < 1e-4
    (H.abs() ** 2).sum().backward()
    for name, p_ in net.named_parameters():
        key = "grad_" + name
        if key in fx:
            assert p_.grad is not None, name
            within(rel_err(p_.grad.detach().cpu(), fx[key]), 1e-4, ("L543",) + tuple((name, rel_err(p_.grad.detach().cpu(), fx[key]))))
    out = net.get_param_dict_inference(batch)
    assert out["output_biquad_coeffs"].shape[-2:] == (11, 6)


def test_f7_front_end():
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset
    fx = load("f7_front_end.npz")
    rirs = fx["rirs"].copy()
    ds = RoomDataset(2, float(fx["fs"]), fx["src"], fx["pos"], rirs, np.array([[0.2, 0.4]]),
                     nfft=int(fx["nfft"]), device=DEV)
    md = MultiRIRDataset(DEV, ds)
    batch = md.collate(range(len(md)))
    for k in ("z_values", "listener_position", "norm_listener_position", "target_early_response",
              "target_late_response", "target_rir_response"):
        assert rel_err(batch[k].cpu(), fx["batch_" + k]) < 1e-6, k
    assert rel_err(rirs, fx["rirs_after"]) < 1e-12      # in-place fade side effect reproduced


# ---------------------------------------------------------------------------------------------
# full-size checks (BASELINE config 2: N = 16, K = 65 537) against the oracle on a small batch,
# plus size-independent properties
# ---------------------------------------------------------------------------------------------
def _full_size_setup(B=2, G=4, nper=4, seed=3):
    from diffgfdn_amd.config import CouplingMatrixType, DiffGFDNConfig, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    from diffgfdn_amd.synthetic import synthetic_room
    fs, nfft = 32000.0, 131072
    room = synthetic_room(B, G, fs, 64000, seed)
    delays = DiffGFDNConfig(num_delay_lines=G * nper, seed=seed).delay_length_samps
    torch.manual_seed(seed)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    net = DiffGFDNVarReceiverPos(fs, G, delays, DEV, fl, of, use_absorption_filters=False,
                                 common_decay_times=room["common_decay_times"], use_colorless_loss=True)
    full = np.fft.rfft(room["rirs"], n=nfft, axis=-1)
    early = room["rirs"].copy()
    early[:, 640:] = 0
    pos = room["receiver_position"] / np.array([10, 13, 1.5])
    batch = {"z_values": torch.tensor(np.exp(1j * 2 * np.pi * np.fft.rfftfreq(nfft))),
             "listener_position": torch.tensor(room["receiver_position"]),
             "norm_listener_position": torch.tensor(pos),
             "target_early_response": torch.tensor(np.fft.rfft(early, n=nfft, axis=-1)),
             "target_rir_response": torch.tensor(full)}
    return net, batch, room, delays


def test_full_size_forward_and_losses_vs_oracle():
    from diffgfdn_amd.colorless_losses import group_spectral_loss
    from diffgfdn_amd.losses import decay_losses
    net, batch, room, delays = _full_size_setup()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(DEV)
    dbatch = _to_dev(batch)
    H, (Hout, _) = net(dbatch)
    # oracle model from the same state
    lin, norm = [], []
    for i in range(0, 100):
        k = f"output_scalars.mlp.model.{i}.weight"
        if k in sd:
            (lin if sd[k].ndim == 2 else norm).append((sd[k], sd[f"output_scalars.mlp.model.{i}.bias"]))
    p = orc.GridModelParams(32000.0, delays, 4, sd["input_gains"], sd["output_gains"], sd["feedback_loop.M"],
                            sd["feedback_loop.alpha"], room["common_decay_times"], lin, norm, 4)
    with torch.no_grad():
        Ho, (Houto, _) = orc.grid_model_forward(p, batch)
        assert rel_err(H.detach().cpu(), Ho) < TOL
        assert rel_err(Hout.detach().cpu(), Houto) < TOL
        l_edr = orc.edr_loss(batch["target_rir_response"], Ho)
        l_edc = orc.edc_loss(batch["target_rir_response"], Ho, orc.ms_to_samps(1500.0, 32000.0), 640)
    total, edr_v, edc_v = decay_losses(H, dbatch["target_rir_response"], edc_start=640, edc_len=48000 - 640)  # weights 1
    assert abs(edr_v.item() - l_edr.item()) < TOL * abs(l_edr.item())
    assert abs(edc_v.item() - l_edc.item()) < TOL * abs(l_edc.item())
    total.backward()
    assert all(torch.isfinite(q.grad).all() for q in net.parameters() if q.grad is not None)


def test_properties_linearity_and_identity():
    from diffgfdn_amd import hip_ops as ops
    from diffgfdn_amd.losses import decay_losses
    torch.manual_seed(0)
    K = 65537
    X1 = torch.randn(2, K, dtype=torch.complex64, device=DEV)
    X2 = torch.randn(2, K, dtype=torch.complex64, device=DEV)
    a = ops.irfft_odd_fwd(X1, K)
    b = ops.irfft_odd_fwd(X2, K)
    c = ops.irfft_odd_fwd(2.0 * X1 - 3.0 * X2, K)
    assert rel_err(c.cpu(), (2.0 * a - 3.0 * b).cpu()) < 1e-5                 # linearity
    # bins above (K-1)/2 do not influence the result (the reference quirk)
    X3 = X1.clone()
    X3[:, (K - 1) // 2 + 1:] = 0
    assert torch.equal(ops.irfft_odd_fwd(X3, K), a)
    # adjoint identity <irfft(X), g> == Re<X, irfft^H(g)>
    g = torch.randn(2, K, device=DEV)
    lhs = (a * g).sum().item()
    gX = ops.irfft_odd_bwd(g, K, K)
    rhs = (X1.real * gX.real + X1.imag * gX.imag).sum().item()
    assert abs(lhs - rhs) < 1e-4 * abs(lhs)
    # identical target and achieved responses give zero decay losses
    H = X1.clone().requires_grad_(True)
    total, e1, e2 = decay_losses(H, X1.clone(), edc_start=640, edc_len=40000)
    assert e1.item() < 1e-6 and e2.item() < 1e-4
    # determinism: the same call twice is bitwise identical
    t1, _, _ = decay_losses(X2.clone().requires_grad_(True), X1, edc_start=640, edc_len=40000)
    t2, _, _ = decay_losses(X2.clone().requires_grad_(True), X1, edc_start=640, edc_len=40000)
    assert t1.item() == t2.item()


def test_errors_are_loud():
    from diffgfdn_amd import hip_ops as ops
    with pytest.raises(RuntimeError):
        ops.irfft_odd_fwd(torch.randn(2, 10, dtype=torch.complex64), 9)       # CPU tensor
    with pytest.raises(RuntimeError):
        ops.stft_power(torch.randn(1, 100, device=DEV), 4096)                 # too short
    with pytest.raises(RuntimeError):
        ops.solve_fwd(torch.zeros(4, dtype=torch.float64, device=DEV), None,
                      torch.zeros(1, 40, 40, device=DEV), torch.zeros(40, device=DEV),
                      torch.ones(40, device=DEV), torch.ones(40, device=DEV))  # block > 32


def test_directional_trainer_step():
    """DirectionalFDNVarReceiverPosTrainer: fused step loss == reference loss of F6, one Adam step runs."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig, TrainerConfig
    from diffgfdn_amd.model import DiffDirectionalFDNVarReceiverPos
    from diffgfdn_amd.trainer import DirectionalFDNVarReceiverPosTrainer
    fx = load("f6_directional.npz")
    fs = float(fx["fs"])
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=1, num_neurons_per_layer=8, num_fourier_features=3)
    net = DiffDirectionalFDNVarReceiverPos(fs, int(fx["G"]), fx["delays"].tolist(), DEV, fl, of,
                                           ambi_order=int(fx["order"]), common_decay_times=fx["T60"][None, :],
                                           use_colorless_loss=True, analysis_matrix=fx["analysis_matrix"])
    net.load_state_dict(_state(fx), strict=True)
    net = net.to(DEV)
    tc = TrainerConfig(use_colorless_loss=False, edc_loss_weight=1.0, train_dir="/tmp/gfdn_t", ir_dir="/tmp/gfdn_a",
                       device="cuda")
    tr = DirectionalFDNVarReceiverPosTrainer(net, tc)
    # the reference fixture used a shorter EDC window and explicit envelopes
    from diffgfdn_amd.losses import directional_edc_loss
    tr.criterion[0] = directional_edc_loss(fx["T60"][None, :], float(fx["edc_len_ms"]), fs,
                                           envelopes=torch.tensor(fx["envelopes"]))
    batch = _to_dev(batch_from(fx))
    batch["target_common_slope_amps"] = torch.tensor(fx["amps"]).to(DEV)
    losses = tr._step_losses(batch)
    assert abs(float(losses["edc_loss"]) - float(fx["loss"])) < TOL * abs(float(fx["loss"]))
    before = net.output_gains.detach().clone()
    total, parts = tr.train_step(batch)
    assert torch.isfinite(total) and not torch.equal(before, net.output_gains.detach())
    # IR export: one wav per receiver with the SH channels (reference :868-921)
    import tempfile
    from scipy.io import wavfile
    with tempfile.TemporaryDirectory() as d:
        tr.save_ir(batch, d, batch["source_position"], batch["listener_position"], norm=False)
        rec = batch["listener_position"].cpu()
        fs_w, data = wavfile.read(f'{d}/ir_({rec[0, 0]:.2f}, {rec[0, 1]:.2f}, {rec[0, 2]:.2f}).wav')
        nch = (int(fx["order"]) + 1) ** 2
        assert fs_w == int(fs) and data.shape == (2 * (batch["z_values"].numel() - 1), nch)


def test_single_pos_trainer_step_matches_oracle():
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig, TrainerConfig
    from diffgfdn_amd.model import DiffGFDNSinglePos
    from diffgfdn_amd.trainer import SinglePosTrainer
    fx = load("f5_single_pos.npz")
    fs = float(fx["fs"])
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=False)
    net = DiffGFDNSinglePos(fs, int(fx["G"]), fx["delays"].tolist(), DEV, fl, OutputFilterConfig(use_svfs=False),
                            use_absorption_filters=False, common_decay_times=fx["T60"][None, :],
                            use_colorless_loss=True)
    net.load_state_dict(_state(fx), strict=True)
    net = net.to(DEV)
    tc = TrainerConfig(use_colorless_loss=True, edc_loss_weight=10.0, sparsity_loss_weight=2.0,
                       train_dir="/tmp/gfdn_t", ir_dir="/tmp/gfdn_a", device="cuda")
    tr = SinglePosTrainer(net, tc, stft_win=256)
    x = {"z_values": torch.tensor(fx["z"]).to(DEV), "target_early_response": torch.tensor(fx["early"]).to(DEV),
         "target_rir_response": torch.tensor(fx["target"]).to(DEV)}
    losses = tr._step_losses(x)
    H = torch.tensor(fx["H"])
    tgt = torch.tensor(fx["target"])
    l_edr = orc.edr_loss(tgt, H, 256, 128)
    l_edc = orc.edc_loss(tgt, H, orc.ms_to_samps(float(np.max(fx["T60"])) * 1e3, fs), orc.ms_to_samps(20.0, fs))
    assert abs(float(losses["edr_loss"]) - l_edr.item()) < TOL * abs(l_edr.item())
    assert abs(float(losses["edc_loss"]) - 10.0 * l_edc.item()) < TOL * abs(10.0 * l_edc.item())
    total, _ = tr.train_step(x)
    assert torch.isfinite(total)
    import tempfile
    from scipy.io import wavfile
    with tempfile.TemporaryDirectory() as d:                      # IR export (reference :664-684)
        h = tr.save_ir(x, d, filename_prefix="single", norm=True)
        fs_w, data = wavfile.read(d + "/single.wav")
        assert data.shape == (h.numel(), 2) and abs(float(np.abs(data).max()) - 1.0) < 1e-6
        assert np.array_equal(data[:, 0], data[:, 1])


def test_rfft_front_end_kernel_and_sh_mix():
    from diffgfdn_amd import hip_ops as ops
    torch.manual_seed(0)
    x = torch.randn(3, 64000, dtype=torch.float64)
    X = ops.rfft_pow2(x.float().to(DEV), 131072)
    ref = torch.fft.rfft(x, n=131072)
    assert rel_err(X.cpu(), ref) < 5e-6
    A = torch.randn(5, 9)
    H = torch.randn(2, 9, 300, dtype=torch.complex64)
    out = ops.sh_to_directional(A.to(DEV), H.to(DEV))
    assert rel_err(out.cpu(), torch.einsum('jl,blk->bjk', A.to(torch.complex64), H)) < 1e-5
    g = torch.randn(2, 5, 300, dtype=torch.complex64)
    back = ops.sh_to_directional(A.to(DEV), g.to(DEV), adjoint=True)
    assert rel_err(back.cpu(), torch.einsum('jl,bjk->blk', A.to(torch.complex64), g)) < 1e-5


def test_band_recombination_filter():
    """full_convolve == scipy.signal.fftconvolve(h, taps, 'full') (reference run_subband_training_treble.py:321-324)."""
    from scipy.signal import fftconvolve
    from diffgfdn_amd.subband import full_convolve
    rng = np.random.RandomState(0)
    h = rng.randn(3, 8192)
    taps = rng.randn(301) * np.hanning(301)
    y = full_convolve(torch.tensor(h, dtype=torch.float32, device=DEV), torch.tensor(taps, dtype=torch.float32))
    ref = np.stack([fftconvolve(h[i], taps, mode="full") for i in range(3)])
    assert y.shape == ref.shape
    assert rel_err(y.cpu(), ref) < 1e-5


def test_split_graph_path_used_for_data_parallel():
    """The N > 1 step replays TWO graphs with the (eager) all-reduce of the flat gradient buffer in
    between.  Exercised here on one GPU with a stand-in all-reduce that doubles the gradients (what
    a 2-rank sum of identical shards would give): parameters must follow Adam on 2 x grad, and must
    equal the single-graph path when the stand-in is the identity."""
    from diffgfdn_amd.config import TrainerConfig
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset
    from diffgfdn_amd.synthetic import synthetic_room
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    fx = load("f234_n12_k257.npz")
    room = synthetic_room(8, 3, 2000.0, 400, seed=5)
    ds = MultiRIRDataset(DEV, RoomDataset(3, 2000.0, room["source_position"], room["receiver_position"],
                                          room["rirs"], room["common_decay_times"], nfft=512, device=DEV))
    tc = TrainerConfig(batch_size=4, num_freq_bins=512, lr=1e-3, io_lr=1e-2, use_colorless_loss=True,
                       use_asym_spectral_loss=True, edc_loss_weight=10.0, sparsity_loss_weight=2.0,
                       use_edc_mask=False, train_dir="/tmp/gfdn_t", ir_dir="/tmp/gfdn_a", device="cuda")
    out = {}
    for mode in ("single", "split_identity"):
        net = _grid_model(fx)
        tr = VarReceiverPosTrainer(net, tc, stft_win=64, capturable=True)
        if mode != "single":
            tr._allreduce = lambda: None                      # identity "all-reduce" -> two graphs
        step = tr.graphed(ds, 4).capture([0, 1, 2, 3])
        assert (step.graph_b is not None) == (mode != "single")
        vals = [float(step(sel)["_total"]) for sel in ([0, 1, 2, 3], [4, 5, 6, 7])]
        out[mode] = (vals, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    assert out["single"][0] == out["split_identity"][0]
    for k, v in out["single"][1].items():
        assert torch.equal(v, out["split_identity"][1][k]), k


@pytest.mark.parametrize("tag", ["zc", "cp"])
def test_f8_source_receiver_model(tag):
    """DiffGFDNVarSourceReceiverPos (source AND receiver gain networks) vs the reference's forward and the
    gradients of sum |H|^2 w.r.t. every parameter (fixture F8; model.py:303-452)."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNVarSourceReceiverPos
    fx = load(f"f8_source_receiver_{tag}.npz")
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=bool(fx["zero_coupling"]))
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    net = DiffGFDNVarSourceReceiverPos(float(fx["fs"]), int(fx["G"]), fx["delays"].tolist(), DEV, fl, of, of,
                                       use_absorption_filters=False, learn_common_decay_times=False,
                                       common_decay_times=fx["T60"][None, :], use_colorless_loss=True)
    net.load_state_dict(_state(fx), strict=True)
    net = net.to(DEV)
    batch = _to_dev(batch_from(fx))
    H, (Hout, _) = net(batch)
    assert rel_err(H.detach().cpu().numpy(), fx["H"]) < TOL
    assert rel_err(Hout.detach().cpu().numpy(), fx["Hout"]) < TOL
    (H.abs() ** 2).sum().backward()
    for name_, prm in net.named_parameters():
        ref = fx["grad_" + name_]
        got = prm.grad.cpu().numpy()
        within(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30), 1e-4, ("L818",) + tuple((name_,)))


def test_f9_colorless_fdn_prototype(tmp_path):
    """ColorlessFDN + ColorlessFDNTrainer vs the reference (fixture F9): forward, training / validation loss,
    gradients, the constructor's energy normalisation; then a short training run that must lower the loss."""
    from diffgfdn_amd.colorless_fdn import ColorlessFDN, ColorlessFDNDataset, ColorlessFDNTrainer
    from diffgfdn_amd.colorless_losses import amse_loss, sparsity_loss
    from diffgfdn_amd.config import TrainerConfig
    fx = load("f9_colorless_fdn.npz")
    net = ColorlessFDN(float(fx["fs"]), fx["delays"].tolist(), DEV, nominal_t60=10.0)
    net.load_state_dict(_state(fx), strict=True)
    net = net.to(DEV)
    z = torch.tensor(fx["z"]).to(DEV)
    H, Hpd = net(z)
    # The reference's prototype runs on a COMPLEX64 grid (colorless_fdn/dataloader.py) and evaluates z ** m in
    # complex64 (feedback_loop.py:330): a phase error of ~m eps32 (1e-4 in the entries of D for delays of a few
    # hundred samples), which the per-bin systems (condition number ~80 at T60 = 10 s) amplify to 6e-4 of the peak
    # response -- noise of the reference itself: the fixture sits 6.1e-4 from a float64 evaluation of the same
    # formula.  Here z^m comes from the exactly reduced phase, so the comparison with the fixture carries PTOL and
    # the comparison with the float64 evaluation (below) the usual tolerance.
    PTOL = 2e-3
    within(rel_err(H.detach().cpu().numpy(), fx["H"]), PTOL, "L840")
    within(rel_err(Hpd.detach().cpu().numpy(), fx["Hpd"]), PTOL, "L841")
    fl = net.feedback_loop
    with torch.no_grad():                                      # float64 evaluation of c^T (D / gamma - Q)^-1 b
        z64 = z.to(torch.complex128)
        D = torch.diag_embed(z64[:, None] ** fl.delays.double()[None, :] / fl.current_gains().double().to(DEV)[None, :])
        Q64 = fl.ortho_param(fl.random_feedback_matrix).double().to(torch.complex128)
        y64 = torch.linalg.solve(D - Q64[None], net.input_gains.double().to(torch.complex128).reshape(1, -1, 1)
                                 .expand(len(z), -1, 1)).squeeze(-1)
        H64 = (y64 * net.output_gains.double().reshape(1, -1)).sum(-1)
    assert rel_err(H.detach().cpu().numpy(), H64.cpu().numpy()) < 1e-5
    assert 3e-4 < rel_err(fx["H"], H64.cpu().numpy()) < 1e-3       # the reference's own complex64 z ** m
    ones = torch.ones(len(z), device=DEV)
    loss = amse_loss()(H, ones) + 1.5 * sparsity_loss()(fl.ortho_param(fl.random_feedback_matrix))
    assert abs(loss.item() - float(fx["loss"])) < 1e-3 * abs(float(fx["loss"]))
    loss.backward()
    for name_, prm in net.named_parameters():
        ref = fx["grad_" + name_]
        within(np.abs(prm.grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30), 3e-3, ("L858",) + tuple((name_,)))
    vloss = amse_loss()(H, ones) + amse_loss()(Hpd, torch.ones_like(Hpd)) \
        + 1.5 * sparsity_loss()(fl.ortho_param(fl.random_feedback_matrix))
    assert abs(vloss.item() - float(fx["valid_loss"])) < 1e-3 * abs(float(fx["valid_loss"]))
    # trainer: normalisation on a 600-point grid as the fixture, then training lowers the loss
    tc = TrainerConfig(device="cuda", use_asym_spectral_loss=True, train_dir=str(tmp_path) + "/")
    net2 = ColorlessFDN(float(fx["fs"]), fx["delays"].tolist(), DEV, nominal_t60=10.0)
    net2.load_state_dict(_state(fx), strict=True)
    net2 = net2.to(DEV)
    tr = ColorlessFDNTrainer(net2, tc, alpha=1.5, lr=0.01, max_epochs=3, batch_size=600)
    within(rel_err(net2.input_gains.detach().cpu().numpy(), fx["norm_input_gains"]), 3e-4, "L868")
    within(rel_err(net2.output_gains.detach().cpu().numpy(), fx["norm_output_gains"]), 3e-4, "L869")
    from scipy.io import wavfile
    h = tr.save_ir(str(tmp_path / "ir"))
    fs_w, data = wavfile.read(str(tmp_path / "ir" / "colorless_fdn_ir.wav"))
    assert fs_w == int(fx["fs"]) and data.dtype == np.float32 and data.shape == (h.numel(), 2)
    assert np.array_equal(data[:, 0], h.float().cpu().numpy())
    ds = ColorlessFDNDataset(1200, DEV)
    batches = [(ds.input[i:i + 300], ds.labels[i:i + 300]) for i in range(0, 1200, 300)]
    tr.train(batches[:3], batches[3:])
    assert tr.train_loss[-1] < tr.train_loss[0]
    assert (tmp_path / "colorless-fdn" / "checkpoints" / "model_e2.pt").exists()


def test_f10_absorption_filters_model():
    """use_absorption_filters: per-line absorption FILTERS Gamma_i(z) inside the per-bin solve (fixture F10: the
    reference designs the GEQ sections, we take its coefficients as data): explicit inverse, forward, gradients."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    fx = load("f10_absorption_filters.npz")
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=False)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    coeffs = torch.tensor(fx["sd_delay_filters"])
    net = DiffGFDNVarReceiverPos(float(fx["fs"]), int(fx["G"]), fx["delays"].tolist(), DEV, fl, of,
                                 use_absorption_filters=True, common_decay_times=fx["T60"],
                                 band_centre_hz=fx["band_centre_hz"].tolist(), use_colorless_loss=True,
                                 absorption_filter_coeffs=coeffs)
    net.load_state_dict(_state(fx), strict=True)
    net = net.to(DEV)
    batch = _to_dev(batch_from(fx))
    P = net.feedback_loop(batch["z_values"][:64].contiguous())
    assert rel_err(P.detach().cpu().numpy(), fx["P_small"]) < TOL
    H, (Hout, _) = net(batch)
    assert rel_err(H.detach().cpu().numpy(), fx["H"]) < TOL
    assert rel_err(Hout.detach().cpu().numpy(), fx["Hout"]) < TOL
    (H.abs() ** 2).sum().backward()
    for name_, prm in net.named_parameters():
        ref = fx["grad_" + name_]
        within(np.abs(prm.grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30), 1e-4, ("L906",) + tuple((name_,)))
    with pytest.raises(NotImplementedError):
        DiffGFDNVarReceiverPos(float(fx["fs"]), int(fx["G"]), fx["delays"].tolist(), DEV, fl, of,
                               use_absorption_filters=True, common_decay_times=fx["T60"])


def test_f11_svf_filters():
    """SVF output filters from an MLP on the grid model and learnable SVF input / output filters on the
    single-position model (fixture F11; gain_filters.py:262-402, model.py:544-592, :723-778, :838-911)."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNSinglePos, DiffGFDNVarReceiverPos
    fx = load("f11_svf_filters.npz")
    fs, G, delays = float(fx["fs"]), int(fx["G"]), fx["delays"].tolist()
    batch = _to_dev(batch_from(fx))
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=True, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4,
                            compress_pole_factor=0.98)
    net = DiffGFDNVarReceiverPos(fs, G, delays, DEV, fl, of, use_absorption_filters=False,
                                 common_decay_times=fx["T60"][None, :], use_colorless_loss=True)
    net.load_state_dict(_state(fx, "gsd_"), strict=True)
    net = net.to(DEV)
    Co = net.output_filters.group_responses(batch)
    assert rel_err(Co.detach().cpu().numpy(), fx["grid_Co"]) < 2e-5
    H, (Hout, _) = net(batch)
    assert rel_err(H.detach().cpu().numpy(), fx["grid_H"]) < TOL
    (H.abs() ** 2).sum().backward()
    for name_, prm in net.named_parameters():
        ref = fx["ggrad_" + name_]
        within(np.abs(prm.grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30), 1e-4, ("L934",) + tuple((name_,)))
    # single position: SVF cascades on both sides, coupled feedback matrix
    fl2 = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=False)
    of2 = OutputFilterConfig(use_svfs=True, compress_pole_factor=1.0)
    sp = DiffGFDNSinglePos(fs, G, delays, DEV, fl2, of2, use_absorption_filters=False,
                           common_decay_times=fx["T60"][None, :], use_colorless_loss=True, input_filter_config=of2)
    sp.load_state_dict(_state(fx, "ssd_"), strict=True)
    sp = sp.to(DEV)
    x = {"z_values": batch["z_values"], "target_early_response": batch["target_early_response"][0].clone()}
    Hs, (Hsout, _) = sp(x)
    assert rel_err(Hs.detach().cpu().numpy(), fx["sp_H"]) < TOL
    assert rel_err(Hsout.detach().cpu().numpy(), fx["sp_Hout"]) < TOL
    (Hs.abs() ** 2).sum().backward()            # the reference cannot back-propagate here (see the fixture script)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in sp.parameters())


def test_svf_grid_trainer_steps():
    """VarReceiverPosTrainer on a grid model with SVF output filters: normalize + a few optimiser steps run on the
    module-forward path and lower the loss."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig, TrainerConfig
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    fx = load("f11_svf_filters.npz")
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=True, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4,
                            compress_pole_factor=0.98)
    net = DiffGFDNVarReceiverPos(float(fx["fs"]), int(fx["G"]), fx["delays"].tolist(), DEV, fl, of,
                                 use_absorption_filters=False, common_decay_times=fx["T60"][None, :],
                                 use_colorless_loss=True)
    net.load_state_dict(_state(fx, "gsd_"), strict=True)
    net = net.to(DEV)
    tc = TrainerConfig(batch_size=3, num_freq_bins=int(fx["nfft"]), lr=1e-2, io_lr=1e-2, use_colorless_loss=True,
                       use_asym_spectral_loss=True, edc_loss_weight=1.0, sparsity_loss_weight=1.0,
                       use_edc_mask=False, train_dir="/tmp/gfdn_t", ir_dir="/tmp/gfdn_a", device="cuda")
    tr = VarReceiverPosTrainer(net, tc, stft_win=64)
    batch = _to_dev(batch_from(fx))
    tr.normalize(batch)
    vals = [float(tr.train_step(batch)[0]) for _ in range(8)]
    assert all(np.isfinite(vals)) and vals[-1] < vals[0]


@pytest.mark.parametrize("G,nper,K", [(2, 2, 33), (3, 4, 257), (3, 5, 100), (4, 4, 1025), (8, 4, 77), (4, 4, 20001),
                                      (2, 4, 9000)])
def test_filter_coupling_solve_kernels(G, nper, K):
    """gfdn_solve_phi_fwd / _bwd (frequency-dependent feedback matrix A(z_k) = BM o kron(Phi_k, 1)) against
    torch.linalg.solve in complex128 and its autograd gradients."""
    from diffgfdn_amd.functional import FrequencyGrid, ResolventSolveFilter
    g = torch.Generator().manual_seed(100 * G + nper)
    N = G * nper
    Q, _ = torch.linalg.qr(torch.randn(N, N, generator=g, dtype=torch.float64))
    BM = Q.to(torch.float32).to(DEV).requires_grad_(True)
    Phi = (torch.randn(K, G, G, 2, generator=g) * 0.5)
    Phi = torch.view_as_complex(Phi).to(DEV).requires_grad_(True)
    delays = torch.randint(20, 90, (N,), generator=g).to(torch.float32).to(DEV)
    ig = (1.0 + 0.2 * torch.rand(N, generator=g)).to(DEV).requires_grad_(True)
    b = torch.randn(N, generator=g).to(DEV).requires_grad_(True)
    z = torch.exp(1j * np.pi * torch.arange(K, dtype=torch.float64) / (K - 1)).to(torch.complex128).to(DEV)
    wgt = torch.view_as_complex(torch.randn(K, N, 2, generator=g)).to(DEV)
    Y = ResolventSolveFilter.apply(BM, Phi, ig, b, FrequencyGrid.of(z), delays, nper)
    (Y * wgt).real.sum().backward()
    got = [BM.grad.clone(), Phi.grad.clone(), ig.grad.clone(), b.grad.clone()]
    # float64 torch reference
    BMr, Phir, igr, br = [t.detach().to(torch.complex128 if t.is_complex() else torch.float64).requires_grad_(True)
                          for t in (BM, Phi, ig, b)]
    A = BMr[None].to(torch.complex128) * Phir.repeat_interleave(nper, 1).repeat_interleave(nper, 2)
    D = torch.diag_embed(z[:, None] ** delays.to(torch.float64)[None, :] * igr[None, :])
    Yr = torch.linalg.solve(D - A, br.to(torch.complex128)[None, :, None].expand(K, N, 1)).squeeze(-1)
    assert rel_err(Y.detach().cpu(), Yr.detach().cpu()) < TOL
    (Yr * wgt.to(torch.complex128)).real.sum().backward()
    for a, r in zip(got, (BMr, Phir, igr, br)):
        within(rel_err(a.cpu(), r.grad.cpu()), 1e-4, "L1004")


def test_f12_filter_coupling():
    """Paraunitary FILTER coupling (fixture F12; feedback_loop.py:90-143, :311-323, :362-373, :413-455): coupling
    polynomial, polynomial feedback matrix, explicit inverse and gradients of the loop; transfer function and
    gradients of the grid model."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.feedback_loop import FeedbackLoop
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    fx = load("f12_filter_coupling.npz")
    M = fx["loop_M"]
    G, n, _ = M.shape
    loop = FeedbackLoop(float(fx["fs"]), G, n, torch.tensor(fx["loop_delays"], dtype=torch.float32), False,
                        coupling_matrix_type=CouplingMatrixType.FILTER, coupling_matrix_order=int(fx["order"]),
                        gains=torch.tensor(fx["loop_gamma"]))
    with torch.no_grad():
        loop.M.copy_(torch.tensor(M))
        loop.unit_vectors.copy_(torch.tensor(fx["loop_unit_vectors"]))
        loop.unitary_matrix.copy_(torch.tensor(fx["loop_unitary_matrix"]))
    loop = loop.to(DEV)
    P = loop(torch.tensor(fx["z"]).to(DEV))
    assert rel_err(loop.phi.detach().cpu(), fx["loop_phi"]) < 1e-5
    assert rel_err(loop.coupled_feedback_matrix.detach().cpu(), fx["loop_A"]) < 1e-5
    assert rel_err(P.detach().cpu(), fx["loop_P"]) < TOL
    (P.abs() ** 2).sum().backward()
    within(rel_err(loop.M.grad.cpu(), fx["loop_grad_M"]), 3e-4, "L1030")
    within(rel_err(loop.unit_vectors.grad.cpu(), fx["loop_grad_unit_vectors"]), 1e-4, "L1031")
    within(rel_err(loop.unitary_matrix.grad.cpu(), fx["loop_grad_unitary_matrix"]), 3e-4, "L1032")
    # grid model
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.FILTER, pu_matrix_order=int(fx["net_order"]))
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    net = DiffGFDNVarReceiverPos(float(fx["fs"]), int(fx["net_G"]), fx["net_delays"].tolist(), DEV, fl, of,
                                 use_absorption_filters=False, common_decay_times=fx["net_T60"][None, :],
                                 use_colorless_loss=True)
    net.load_state_dict(_state(fx, "net_sd_"), strict=True)
    net = net.to(DEV)
    batch = _to_dev(batch_from(fx, "net_batch_"))
    H, (Hout, _) = net(batch)
    assert rel_err(H.detach().cpu(), fx["net_H"]) < TOL
    assert rel_err(Hout.detach().cpu(), fx["net_Hout"]) < TOL
    (H.abs() ** 2).sum().backward()
    for name_, prm in net.named_parameters():
        ref = fx["net_grad_" + name_]
        within(np.abs(prm.grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30), 1e-4, ("L1048",) + tuple((name_,)))


def test_f15_full_band_svf_with_absorption_filters():
    """The full-band configuration as the reference runs it (fixture F15a; data/config/treble_data_grid_training_full_
    band_colorless_loss.yml:5-6, :22-26): SVF output filters from the 10 x 64 network on 20 Fourier features TOGETHER
    with absorption filters on the delay lines, N = 12 = 3 x 4 -- forward, sub-FDN output and every parameter
    gradient against the reference (gain_filters.py:262-402, feedback_loop.py:332-344, :376-381, model.py:544-619)."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    fx = load("f15_full_band.npz")
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR,
                            use_zero_coupling=bool(fx["fb_use_zero_coupling"]))
    of = OutputFilterConfig(use_svfs=True, num_hidden_layers=10, num_neurons_per_layer=64, num_fourier_features=20,
                            compress_pole_factor=float(fx["fb_compress_pole_factor"]))
    coeffs = torch.tensor(fx["fb_sd_delay_filters"])
    net = DiffGFDNVarReceiverPos(float(fx["fs"]), int(fx["G"]), fx["delays"].tolist(), DEV, fl, of,
                                 use_absorption_filters=True, common_decay_times=fx["T60"],
                                 band_centre_hz=fx["band_centre_hz"].tolist(), use_colorless_loss=True,
                                 absorption_filter_coeffs=coeffs)
    net.load_state_dict(_state(fx, "fb_sd_"), strict=True)
    net = net.to(DEV)
    batch = _to_dev(batch_from(fx))
    H, (Hout, _) = net(batch)
    assert rel_err(H.detach().cpu().numpy(), fx["fb_H"]) < TOL
    assert rel_err(Hout.detach().cpu().numpy(), fx["fb_Hout"]) < TOL
    (H.abs() ** 2).sum().backward()
    worst = {}
    for name_, prm in net.named_parameters():
        ref = fx["fb_grad_" + name_]
        worst[name_] = np.abs(prm.grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30)
        within(worst[name_], 1e-4, ("L1079",) + tuple((name_, worst[name_])))


def test_f15_filter_coupling_with_absorption_filters():
    """FILTER coupling TOGETHER with absorption filters (fixture F15b; the reference's forward handles both in one pass,
    feedback_loop.py:362-386): explicit inverse on a few bins, forward and every gradient of the grid model, through
    gfdn_solve_phi_absorb_fwd / _bwd (per-bin complex 1 / Gamma_i(z_k) on the diagonal of the complex-A systems)."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    fx = load("f15_full_band.npz")
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.FILTER, pu_matrix_order=int(fx["fa_order"]))
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    coeffs = torch.tensor(fx["fa_sd_delay_filters"])
    net = DiffGFDNVarReceiverPos(float(fx["fs"]), int(fx["G"]), fx["delays"].tolist(), DEV, fl, of,
                                 use_absorption_filters=True, common_decay_times=fx["T60"],
                                 band_centre_hz=fx["band_centre_hz"].tolist(), use_colorless_loss=True,
                                 absorption_filter_coeffs=coeffs)
    net.load_state_dict(_state(fx, "fa_sd_"), strict=True)
    net = net.to(DEV)
    batch = _to_dev(batch_from(fx))
    P = net.feedback_loop(batch["z_values"][:48].contiguous())
    assert rel_err(P.detach().cpu().numpy(), fx["fa_P_small"]) < TOL
    H, (Hout, _) = net(batch)
    assert rel_err(H.detach().cpu().numpy(), fx["fa_H"]) < TOL
    assert rel_err(Hout.detach().cpu().numpy(), fx["fa_Hout"]) < TOL
    (H.abs() ** 2).sum().backward()
    for name_, prm in net.named_parameters():
        ref = fx["fa_grad_" + name_]
        within(np.abs(prm.grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30), 1e-4, ("L1107",) + tuple((name_,)))


def test_filter_coupling_trainer_steps():
    """VarReceiverPosTrainer on a grid model with paraunitary FILTER coupling: normalize + a few optimiser steps lower
    the loss and move the coupling parameters (unit_vectors, unitary_matrix ride in the default lr group)."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig, TrainerConfig
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    fx = load("f12_filter_coupling.npz")
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.FILTER, pu_matrix_order=int(fx["net_order"]))
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    net = DiffGFDNVarReceiverPos(float(fx["fs"]), int(fx["net_G"]), fx["net_delays"].tolist(), DEV, fl, of,
                                 use_absorption_filters=False, common_decay_times=fx["net_T60"][None, :],
                                 use_colorless_loss=True)
    net.load_state_dict(_state(fx, "net_sd_"), strict=True)
    net = net.to(DEV)
    uv0 = net.feedback_loop.unit_vectors.detach().clone()
    um0 = net.feedback_loop.unitary_matrix.detach().clone()
    tc = TrainerConfig(batch_size=3, num_freq_bins=int(fx["net_nfft"]), lr=1e-2, io_lr=1e-2, use_colorless_loss=True,
                       use_asym_spectral_loss=True, edc_loss_weight=1.0, sparsity_loss_weight=1.0,
                       use_edc_mask=False, train_dir="/tmp/gfdn_t", ir_dir="/tmp/gfdn_a", device="cuda")
    tr = VarReceiverPosTrainer(net, tc, stft_win=64)
    batch = _to_dev(batch_from(fx, "net_batch_"))
    tr.normalize(batch)
    vals = [float(tr.train_step(batch)[0]) for _ in range(8)]
    assert all(np.isfinite(vals)) and vals[-1] < vals[0]
    assert float((net.feedback_loop.unit_vectors.detach() - uv0).abs().max()) > 0
    assert float((net.feedback_loop.unitary_matrix.detach() - um0).abs().max()) > 0


@pytest.mark.parametrize("N,K", [(8, 1025), (16, 513), (24, 300)])
def test_precise_solve(N, K):
    """gfdn_solve_precise_fwd / _bwd (float64 matrix entries, float64 inverse gains, float64 elimination) on nearly
    lossless systems (T60 = 30 s at 48 kHz, delays up to 4000 samples: condition numbers of 1e3..1e5), against
    torch.linalg.solve in complex128 and its autograd gradients; the float32 kernels on the same systems for scale."""
    from diffgfdn_amd.functional import FrequencyGrid, ResolventSolve
    g = torch.Generator().manual_seed(7 * N + K)
    Q, _ = torch.linalg.qr(torch.randn(N, N, generator=g, dtype=torch.float64))
    A = Q.to(torch.float32).to(DEV).unsqueeze(0).requires_grad_(True)
    delays = torch.randint(800, 4000, (N,), generator=g).to(torch.float32).to(DEV)
    gamma = 10.0 ** (-3.0 * delays.double() / (48000.0 * 30.0))
    ig = (1.0 / gamma).requires_grad_(True)                    # float64
    b = torch.randn(N, generator=g).to(DEV).requires_grad_(True)
    z = torch.exp(1j * np.pi * torch.arange(K, dtype=torch.float64) / (K - 1)).to(torch.complex128).to(DEV)
    wgt = torch.view_as_complex(torch.randn(K, N, 2, generator=g)).to(DEV)
    grid = FrequencyGrid.of(z)
    Y = ResolventSolve.apply(A, ig, b, grid, delays, False, None, True)
    (Y * wgt).real.sum().backward()
    got = [A.grad.clone(), ig.grad.clone(), b.grad.clone()]
    Ar, igr, br = [t.detach().double().requires_grad_(True) for t in (A, ig, b)]
    D = torch.diag_embed(z[:, None] ** delays.double()[None, :] * igr[None, :])
    Yr = torch.linalg.solve(D - Ar[0].to(torch.complex128)[None], br.to(torch.complex128)[None, :, None]
                            .expand(K, N, 1)).squeeze(-1)
    assert rel_err(Y.detach().cpu(), Yr.detach().cpu()) < 1e-6
    (Yr * wgt.to(torch.complex128)).real.sum().backward()
    for a, r in zip(got, (Ar, igr, br)):
        assert rel_err(a.cpu(), r.grad.cpu()) < 1e-5
    with torch.no_grad():
        Y32 = ResolventSolve.apply(A.detach(), ig.detach().float(), b.detach(), grid, delays, False)
    assert rel_err(Y32.cpu(), Yr.detach().cpu()) > 10 * rel_err(Y.detach().cpu(), Yr.detach().cpu())


def test_svf_graphed_step_equals_eager_step():
    """Grid model with SVF output filters: the step replayed from a HIP graph (rows batch gathered for the model's
    forward, decay targets from the dataset stores) equals the eager step on collated batches -- same host masks,
    same state; b, c normalised once per epoch as the reference does for full-band models (trainer.py:365-369)."""
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig, TrainerConfig
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    from diffgfdn_amd.synthetic import synthetic_room
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    room = synthetic_room(12, 3, 8000.0, 5000, seed=3)
    ds = MultiRIRDataset(DEV, RoomDataset(3, 8000.0, room["source_position"], room["receiver_position"],
                                          room["rirs"], room["common_decay_times"], nfft=8192, device=DEV))
    tc = TrainerConfig(batch_size=4, num_freq_bins=8192, lr=1e-3, io_lr=1e-2, use_colorless_loss=True,
                       use_asym_spectral_loss=True, edc_loss_weight=10.0, sparsity_loss_weight=2.0, use_edc_mask=True,
                       train_dir="/tmp/gfdn_t", ir_dir="/tmp/gfdn_a", device="cuda")
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=True, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4,
                            compress_pole_factor=0.98)
    delays = [173, 181, 191, 193, 197, 199, 211, 223, 227, 229, 233, 401]
    sels = ([0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11])
    results = []
    for mode in ("eager", "graph"):
        torch.manual_seed(5)
        net = DiffGFDNVarReceiverPos(8000.0, 3, delays, DEV, fl, of, use_absorption_filters=False,
                                     common_decay_times=room["common_decay_times"], use_colorless_loss=True).to(DEV)
        tr = VarReceiverPosTrainer(net, tc, stft_win=512, capturable=(mode == "graph"))
        tr.normalize(ds.collate(sels[0]))
        if mode == "graph":
            step = tr.graphed(ds, 4, mask_source="host").capture(sels[0])
        torch.manual_seed(77)
        tot = []
        for sel in sels:
            if mode == "eager":
                t, _ = tr.train_step(ds.collate(sel))
                tot.append(float(t))
            else:
                tot.append(float(step(sel)["_total"]))
        results.append((tot, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
    (t0, s0), (t1, s1) = results
    for a, b in zip(t0, t1):
        assert abs(a - b) < 1e-5 * abs(a), (t0, t1)
    for k in s0:
        within(rel_err(s1[k], s0[k]), 1e-4, ("L1212",) + tuple((k,)))


def test_save_ir_writes_reference_named_wavs(tmp_path):
    """VarReceiverPosTrainer.save_ir (trainer.py:503-564): h = irfft(H) per receiver as 32-bit float stereo wav files
    with the reference's file names; the samples equal torch.fft.irfft of the returned response."""
    from scipy.io import wavfile
    from diffgfdn_amd.config import TrainerConfig
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    fx = load("f234_n12_k257.npz")
    net = _grid_model(fx)
    tc = TrainerConfig(batch_size=4, num_freq_bins=int(fx["nfft"]), use_colorless_loss=True, train_dir=str(tmp_path / "t"),
                       ir_dir=str(tmp_path / "ir"), device="cuda")
    tr = VarReceiverPosTrainer(net, tc, stft_win=int(fx["win"]))
    batch = _to_dev(batch_from(fx))
    H, Hsub = tr.save_ir(batch, directory=str(tmp_path / "ir"), src_pos=batch["source_position"],
                         rec_pos=batch["listener_position"], norm=False)
    rec = batch["listener_position"].cpu()
    want = torch.fft.irfft(H.detach().cpu().to(torch.complex128)).numpy()
    for r in range(rec.shape[0]):
        name = f'ir_({rec[r, 0]:.2f}, {rec[r, 1]:.2f}, {rec[r, 2]:.2f}).wav'
        fs, data = wavfile.read(str(tmp_path / "ir" / name))
        assert fs == int(fx["fs"]) and data.dtype == np.float32 and data.shape == (want.shape[1], 2)
        assert np.array_equal(data[:, 0], data[:, 1])
        assert rel_err(data[:, 0], want[r]) < 1e-5
    H2, h = tr.save_ir(batch, norm=True)                      # nothing written: (H, h)
    assert abs(float(h.abs().max()) - 1.0) < 1e-6 and rel_err(H2.cpu(), H.cpu()) == 0.0


def test_load_dataset_grid_loaders():
    """load_dataset on a RoomDataset (reference dataloader.py:780-867): train / valid / held-out test loaders over
    disjoint receiver sets, batches with the reference's collate keys, tensors resident on the device."""
    from diffgfdn_amd.dataloader import RoomDataset, get_dataloader, load_dataset
    from diffgfdn_amd.synthetic import synthetic_room
    room = synthetic_room(20, 3, 8000.0, 5000, seed=5)
    rd = RoomDataset(3, 8000.0, room["source_position"], room["receiver_position"], room["rirs"],
                     room["common_decay_times"], nfft=8192, device=DEV)
    torch.manual_seed(3)
    train, valid, test = load_dataset(rd, DEV, train_valid_split_ratio=0.75, batch_size=4, hold_out_test_set=True,
                                      test_set_ratio=0.1, test_set_seed=99)
    sets = [set(l.indices) for l in (train, valid, test)]
    assert sets[0] | sets[1] | sets[2] == set(range(20)) and sum(len(s_) for s_ in sets) == 20
    assert len(sets[2]) == 2 and len(sets[0]) == int(0.75 * 18)
    b = next(iter(train))
    for key in ("z_values", "source_position", "listener_position", "norm_listener_position",
                "target_early_response", "target_late_response", "target_rir_response"):
        assert key in b and b[key].is_cuda
    assert b["target_rir_response"].shape == (4, 4097)
    assert len(get_dataloader(train.dataset, 5, shuffle=False, drop_last=True)) == 4
