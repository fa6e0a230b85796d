"""Band bank on the MI355X (diffgfdn_amd/bandbank.py): all octave bands stepped by one launch per stage.

Parity chain: the band-stacked C entry points against the plain ones on the same data; one bank step
(normalize + forward + losses + backward + Adam) against every band's OWN trainer step and against the
CPU oracle; the graph-replayed bank step against the eager one; per-band checkpoints load back into
the reference-shaped module."""
import numpy as np
import pytest
from tests.margins import within
import torch

from tests.helpers import philox_mask, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
FS, NFFT, WIN = 8000.0, 8192, 512
G, NPER = 3, 4
BANDS = (250.0, 500.0, 1000.0)


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


def _band_filters():
    from scipy.signal import firwin
    out = []
    for f in BANDS:
        taps = firwin(257, [f / np.sqrt(2), f * np.sqrt(2)], pass_zero=False, fs=FS)
        out.append(np.fft.rfft(taps, n=NFFT))
    return np.stack(out)


def _delays(q):
    base = [173, 181, 191, 193, 197, 199, 211, 223, 227, 229, 233, 401]
    return [d + 2 * q for d in base]


def _build_net(q, t60max=0.5):
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    torch.manual_seed(100 + q)
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    T60 = np.linspace(0.2, t60max, G)[None, :]
    return DiffGFDNVarReceiverPos(FS, G, _delays(q), DEV, fl, of, use_absorption_filters=False,
                                  common_decay_times=T60, use_colorless_loss=True).to(DEV)


def _build_data(q, R=12, t60max=0.5):
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset
    from diffgfdn_amd.synthetic import synthetic_room
    room = synthetic_room(R, G, FS, 5000, seed=10 + q, t60_range=(0.2, t60max))
    ds = MultiRIRDataset(DEV, RoomDataset(G, FS, room["source_position"], room["receiver_position"],
                                          room["rirs"].copy(), room["common_decay_times"], nfft=NFFT, device=DEV))
    return room, ds


def _tc(mask=True, **kw):
    from diffgfdn_amd.config import SubbandProcessingConfig, TrainerConfig
    return TrainerConfig(batch_size=4, num_freq_bins=NFFT, lr=1e-3, io_lr=1e-2, use_colorless_loss=True,
                         use_asym_spectral_loss=True, edc_loss_weight=10.0, sparsity_loss_weight=2.0,
                         use_edc_mask=mask, train_dir="/tmp/gfdn_bank/t", ir_dir="/tmp/gfdn_bank/a", device="cuda",
                         subband_process_config=SubbandProcessingConfig(
                             centre_frequency=500.0, frequency_range=(63, 8000), num_fraction_octaves=1), **kw)


# ---------------------------------------------------------------------------------------------
def test_banded_kernels_equal_plain_kernels():
    from diffgfdn_amd import hip_ops as ops
    g = torch.Generator(device="cpu").manual_seed(5)
    nb, K, B, R = 3, 1000, 5, 9
    N = G * NPER
    Y = torch.randn(K, nb * N, 2, generator=g).to(DEV)
    Y = torch.view_as_complex(Y)
    c = torch.randn(nb * N, generator=g).to(DEV)
    rgain = torch.randn(nb * B, G, generator=g).to(DEV)
    direct = torch.view_as_complex(torch.randn(nb * R, K, 2, generator=g).to(DEV))
    rows = torch.tensor([q * R + int(i) for q in range(nb) for i in torch.randperm(R, generator=g)[:B]], device=DEV)
    filt = torch.view_as_complex(torch.randn(nb, K, 2, generator=g).to(DEV))
    gH = torch.view_as_complex(torch.randn(nb * B, K, 2, generator=g).to(DEV))
    H, S = ops.compose_fwd(Y, c, rgain, NPER, direct, filt, want_S=True, direct_rows=rows, nbands=nb)
    gY, gc, grg = ops.compose_bwd(Y, c, rgain, NPER, gH, filt, nbands=nb)
    for q in range(nb):
        Yq = Y[:, q * N:(q + 1) * N].contiguous()
        sl = slice(q * B, (q + 1) * B)
        Hq, Sq = ops.compose_fwd(Yq, c[q * N:(q + 1) * N], rgain[sl], NPER, direct, filt[q], want_S=True,
                                 direct_rows=rows[sl].contiguous())
        assert torch.equal(H[sl], Hq) and torch.equal(S[q * G:(q + 1) * G], Sq)
        gYq, gcq, grgq = ops.compose_bwd(Yq, c[q * N:(q + 1) * N], rgain[sl], NPER, gH[sl].contiguous(), filt[q])
        assert torch.equal(gY[:, q * N:(q + 1) * N], gYq)
        assert torch.equal(gc[q * N:(q + 1) * N], gcq) and torch.equal(grg[sl], grgq)
    # gain network: one packed parameter row per band
    lib_P = ops._lib.load().gfdn_mlp_param_count(4, 16, 2, G)
    w = (0.3 * torch.randn(nb, lib_P, generator=g)).to(DEV)
    pos = torch.rand(nb * R, 3, generator=g, dtype=torch.float64).to(DEV)
    fpi = (torch.exp(torch.linspace(0, np.log(32.0), 4)) * np.pi).to(DEV)
    gains, xhat, rstd = ops.mlp_gains_fwd(pos, fpi, w, 16, 2, G, -1.0, 1.0, rows, nbands=nb)
    gg = torch.randn(nb * B, G, generator=g).to(DEV)
    gw = ops.mlp_gains_bwd(pos, fpi, w, 16, 2, G, -1.0, 1.0, gains, xhat, rstd, gg, rows, nbands=nb)
    for q in range(nb):
        sl = slice(q * B, (q + 1) * B)
        rq = rows[sl].contiguous()
        gq, xq, sq = ops.mlp_gains_fwd(pos, fpi, w[q], 16, 2, G, -1.0, 1.0, rq)
        assert torch.equal(gains[sl], gq) and torch.equal(xhat[sl], xq) and torch.equal(rstd[sl], sq)
        gwq = ops.mlp_gains_bwd(pos, fpi, w[q], 16, 2, G, -1.0, 1.0, gq, xq, sq, gg[sl].contiguous(), rq)
        assert torch.equal(gw[q], gwq)
    # bookkeeping
    lg = torch.rand(nb * G, generator=g).to(DEV)
    Q = torch.randn(nb * G, NPER, NPER, generator=g).to(DEV)
    out, gQ = ops.colorless_terms(lg, Q, 1.0, 2.0, 0.5, nbands=nb)
    a = torch.rand(nb * B, 3, generator=g).to(DEV)
    div = torch.rand(nb * R, generator=g).to(DEV) + 0.5
    b = torch.rand(nb * B, generator=g).to(DEV)
    ws = ops.weighted_sums(a, 1.5, b, 10.0, div, rows, nbands=nb)
    for q in range(nb):
        oq, gQq = ops.colorless_terms(lg[q * G:(q + 1) * G], Q[q * G:(q + 1) * G], 1.0, 2.0, 0.5)
        assert torch.equal(out[q], oq) and torch.equal(gQ[q * G:(q + 1) * G], gQq)
        sl = slice(q * B, (q + 1) * B)
        wq = ops.weighted_sums(a[sl].contiguous(), 1.5, b[sl].contiguous(), 10.0, div, rows[sl].contiguous())
        assert torch.equal(ws[q], wq)


def _bank_setup(mask=True, t60max=None):
    """``t60max``: one longest decay time per band (None: 0.5 s for all) -- bands that differ in it have EDC windows of
    different lengths (reference trainer.py:56-59)."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    t60max = [0.5] * len(BANDS) if t60max is None else list(t60max)
    nets = [_build_net(q, t60max[q]) for q in range(len(BANDS))]
    data = [_build_data(q, t60max=t60max[q]) for q in range(len(BANDS))]
    filt = torch.tensor(_band_filters(), device=DEV).to(torch.complex64)
    bank = BandBank(nets)
    tr = BandBankTrainer(bank, _tc(mask), subband_filter_freq_resp=filt, stft_win=WIN, band_names=BANDS)
    sds = BandStackedDataset([d for _, d in data])
    start, length = tr._decay_window(NFFT // 2 + 1)
    sds.precompute_decay_targets(WIN, *tr._target_window(NFFT // 2 + 1))
    return nets, data, filt, bank, tr, sds, (start, length)


def test_bank_step_equals_band_steps_and_oracle():
    """One bank step == every band's own VarReceiverPosTrainer step (same kernels, same mask), and the
    bank's per-band losses / post-Adam gains match the CPU oracle of the reference step."""
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    from oracle import gfdn_oracle as orc
    from oracle.cpu_trainer import OracleGridTrainer
    nets, data, filt, bank, tr, sds, (start, length) = _bank_setup()
    sels = [[0, 3, 5, 7], [1, 2, 8, 11], [4, 6, 9, 10]]
    mw_np, count = philox_mask(99, 0, length, 1.0 / 4)
    mw = torch.tensor(mw_np, device=DEV)
    sd0 = [{k: v.detach().cpu().clone() for k, v in net.state_dict().items()} for net in nets]

    batch = sds.collate(sds.global_rows(sels))
    tr.normalize(batch)
    tr.optimizer.zero_grad(set_to_none=True)
    losses = tr._step_losses(batch, mask_prenorm=mw, defer_total=True)
    heads = losses.pop("_heads")
    torch.autograd.backward(heads, [torch.ones(len(BANDS), device=DEV)] * 2)
    tr.optimizer.step()
    got = {k: v.detach().cpu().numpy() for k, v in losses.items()}
    assert got["edr_loss"].shape == (len(BANDS),)

    for q in range(len(BANDS)):
        ref_net = _build_net(q)
        ref_net.load_state_dict(sd0[q], strict=True)
        rtr = VarReceiverPosTrainer(ref_net, _tc(True), subband_filter_freq_resp=filt[q], stft_win=WIN,
                                    capturable=True)
        ds = data[q][1]
        ds.precompute_decay_targets(WIN, start, length)
        b = ds.collate(sels[q], lean=True)
        rtr.normalize(b)
        rtr.optimizer.zero_grad(set_to_none=True)
        rl = rtr._step_losses(b, mask_prenorm=mw)
        rl.pop("_total").backward()
        rtr.optimizer.step()
        for k, v in rl.items():
            assert abs(float(v) - got[k][q]) <= 1e-6 * abs(float(v)) + 1e-9, (q, k, float(v), got[k][q])
        for k, v in ref_net.state_dict().items():
            assert rel_err(nets[q].state_dict()[k].detach().cpu(), v.detach().cpu()) < 1e-5, (q, k)

    # CPU oracle of the reference step for every band (float64 restatement pinned to the golden vectors)
    keep = torch.argwhere(torch.tensor(mw_np) > 0)
    for q in range(len(BANDS)):
        room, _ = data[q]
        sd = sd0[q]
        lin, norm = [], []
        for i in range(64):
            k = f"output_scalars.mlp.model.{i}.weight"
            if k in sd:
                (lin if sd[k].ndim == 2 else norm).append((sd[k].clone(), sd[f"output_scalars.mlp.model.{i}.bias"].clone()))
        p = orc.GridModelParams(FS, _delays(q), G, sd["input_gains"].clone(), sd["output_gains"].clone(),
                                sd["feedback_loop.M"].clone(), sd["feedback_loop.alpha"].clone(),
                                np.linspace(0.2, 0.5, G)[None, :], lin, norm, 4)
        otr = OracleGridTrainer(p, lr=1e-3, io_lr=1e-2, edr_weight=1.0, edc_weight=10.0, spectral_weight=1.0,
                                sparsity_weight=2.0, use_asym=True, win=WIN, hop=WIN // 2,
                                subband_filter=filt[q].cpu().to(torch.complex128))
        _, dq = data[q]
        idx = torch.tensor(sels[q])
        # the oracle takes the dataset's own float64 front end (early_response_c128: the float64 transform of the early
        # RIRs, as the reference's dataloader computes it) so that the comparison isolates the step
        ob = {"z_values": dq.z_values.cpu(),
              "norm_listener_position": dq.norm_listener_position[idx].cpu(),
              "listener_position": dq.listener_positions[idx].cpu(),
              "target_early_response": dq.early_response_c128(idx).cpu(),
              "target_rir_response": dq.rir_mag_response[idx].cpu().to(torch.complex128)}
        otr.normalize(ob)
        ototal, oparts = otr.train_step(ob, keep)
        for k, v in oparts.items():
            assert abs(got[k][q] - v) < 1e-4 * abs(v) + 1e-7, (q, k, got[k][q], v)
        for name in ("input_gains", "output_gains"):
            a = getattr(nets[q], name).detach().cpu()
            assert rel_err(a, getattr(p, name).detach()) < 1e-4, (q, name)


def test_graphed_bank_step_equals_eager_bank_step():
    nets, data, filt, bank, tr, sds, (start, length) = _bank_setup()
    sel_steps = [[[0, 3, 5, 7], [1, 2, 8, 11], [4, 6, 9, 10]],
                 [[1, 2, 4, 6], [0, 5, 7, 9], [3, 8, 10, 11]],
                 [[8, 9, 10, 11], [3, 4, 6, 10], [0, 1, 2, 5]]]
    step = tr.graphed(sds, 4, mask_seed=777).capture(sds.global_rows(sel_steps[0]))
    got, got_grad = [], []
    for s in sel_steps:
        got.append(step(sds.global_rows(s))["_total"].detach().cpu().numpy().copy())
        got_grad.append(tr.optimizer.flat_grad.detach().cpu().numpy().copy())
    assert int(step.mask_state.item()) == len(sel_steps)

    nets2, data2, filt2, bank2, tr2, sds2, _ = _bank_setup()
    want = []
    for i, s in enumerate(sel_steps):
        b = sds2.collate(sds2.global_rows(s))
        mw = torch.tensor(philox_mask(777, i, length, 1.0 / 4)[0], device=DEV)
        # normalize inside the step, as the captured graph does
        if tr2._fused is not None:
            losses = tr2._fused.run(b, mw, 1.0, normalize_first=True, train=True, opt_step=False)
            total = losses["_total"]
        else:
            tr2.optimizer.zero_grad(set_to_none=True)
            losses = tr2._step_losses(b, mask_prenorm=mw, defer_total=True, normalize_first=True)
            heads = losses.pop("_heads")
            torch.autograd.backward(heads, [torch.ones(len(BANDS), device=DEV)] * 2)
            tr2.optimizer.pack_grads()
            total = heads[0].detach() + heads[1].detach()
        if i == 0:
            # same state, same inputs: the replayed gradients are the eager ones to the last bit (Adam's
            # first update is sign(g), so the loss comparison below could not see a wrong magnitude)
            assert np.array_equal(tr2.optimizer.flat_grad.cpu().numpy(), got_grad[0])
        tr2.optimizer.step()
        want.append(total.cpu().numpy())
    for a, b in zip(got, want):
        assert np.allclose(a, b, rtol=1e-5, atol=0), (got, want)
    for q in range(len(BANDS)):
        for k, v in nets[q].state_dict().items():
            within(rel_err(v.detach().cpu(), nets2[q].state_dict()[k].detach().cpu()), 1e-6, ("bank L264", q, k))


@pytest.mark.parametrize("S", [2, 4])
def test_pipelined_chain_equals_single_steps(S):
    """run_schedule with ``pipe_steps`` steps per graph (side stream running ahead: the gain network's parameters
    stepped on it, the next step's receivers and gains fetched behind its backward; bankstep.StepPipe) against the same
    batches stepped one graph replay at a time: bit-equal losses of every step and bit-equal parameters / Adam state."""
    _, _, _, _, ta, stacked, _ = _bank_setup(mask=True)
    _, _, _, _, tb, _, _ = _bank_setup(mask=True)
    B = 4
    batches = [[1, 5, 7, 10], [0, 2, 3, 11], [4, 6, 8, 9], [2, 3, 5, 7], [11, 0, 9, 1], [3, 4, 5, 6], [7, 8, 9, 10]]
    rows = [stacked.global_rows([b] * len(BANDS)) for b in batches]
    # (the chained steps form the output stage in the frequency domain; the single steps are held to the same form here,
    # since the time-domain output stage agrees with it to rounding only -- that comparison has its own test)
    ta._fused.linear_transforms = tb._fused.linear_transforms = False
    sa = ta.graphed(stacked, B, mask_seed=5)
    sb = tb.graphed(stacked, B, mask_seed=5)
    sa.pipe_steps, sb.pipe_steps = 0, S
    assert sb._pipe_ok() and not sa._pipe_ok()
    single = [{k: v.clone() for k, v in out.items()} for out in sa.run_schedule(rows)]
    piped = [{k: v.clone() for k, v in out.items()} for out in sb.run_schedule(rows)]
    assert sb.graph_p is not None and len(piped) == len(rows)
    for i, (a, b) in enumerate(zip(single, piped)):
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k)
    for pa, pb in zip(ta.net.parameters(), tb.net.parameters()):
        assert torch.equal(pa, pb)
    for a, b in zip(ta.optimizer.state_tensors(), tb.optimizer.state_tensors()):
        assert torch.equal(a, b)
    assert float(tb.optimizer.step_count.item()) == len(rows) == float(tb.optimizer.step_count2.item())
    # a second schedule on the same graphs (the chain's buffers are refilled by the prologue)
    single = [{k: v.clone() for k, v in out.items()} for out in sa.run_schedule(rows[::-1])]
    piped = [{k: v.clone() for k, v in out.items()} for out in sb.run_schedule(rows[::-1])]
    for i, (a, b) in enumerate(zip(single, piped)):
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k)


def test_scheduled_steps_equal_per_call_steps():
    """load_schedule / run_next (the epoch's batches on the device, every step fetching the next one's receivers
    itself) against step(rows) per call with the host copy in front: same batches -> bit-equal losses and parameters;
    the schedule wraps around and chunks beyond its capacity."""
    _, _, _, _, ta, stacked, _ = _bank_setup(mask=True)
    _, _, _, _, tb, _, _ = _bank_setup(mask=True)
    for pa, pb in zip(ta.net.parameters(), tb.net.parameters()):
        assert torch.equal(pa, pb)
    B = 4
    batches = [[1, 5, 7, 10], [0, 2, 3, 11], [4, 6, 8, 9], [2, 3, 5, 7], [11, 0, 9, 1]]
    rows = [stacked.global_rows([b] * len(BANDS)) for b in batches]
    sa = ta.graphed(stacked, B, mask_seed=99)
    sb = tb.graphed(stacked, B, mask_seed=99)
    sb.sched_cap = 2                                   # 5 steps -> three uploads
    per_call = [{k: v.clone() for k, v in sa(r).items()} for r in rows]
    sched = [{k: v.clone() for k, v in out.items()} for out in sb.run_schedule(rows)]
    for a, b in zip(per_call, sched):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    for pa, pb in zip(ta.net.parameters(), tb.net.parameters()):
        assert torch.equal(pa, pb)
    # wrap-around: a schedule of two batches run three times = batches 0, 1, 0
    sb.load_schedule(rows[:2])
    for r in (rows[0], rows[1], rows[0]):
        a = {k: v.clone() for k, v in sa(r).items()}
        b = sb.run_next()
        for k in a:
            assert torch.equal(a[k], b[k]), k
    with pytest.raises(ValueError):
        sb.load_schedule([rows[0][:-1]])


@pytest.mark.parametrize("mask", [True, False])
def test_fused_bank_step_equals_autograd_bank_step(mask):
    """The explicit launch sequence on the polynomial form of the block transfer functions (bankstep.py) against the
    autograd path on the per-bin elimination kernels: per-band losses, every gradient, the post-Adam state -- and
    the no-grad (validation) form of the same step."""
    from diffgfdn_amd.bandbank import BandBankTrainer
    sels = [[0, 3, 5, 7], [1, 2, 8, 11], [4, 6, 9, 10]]
    res = {}
    for fused in (True, False):
        BandBankTrainer.use_fused = fused
        try:
            nets, data, filt, bank, tr, sds, (start, length) = _bank_setup(mask)
        finally:
            BandBankTrainer.use_fused = True
        assert (tr._fused is not None) == fused
        mw = torch.tensor(philox_mask(99, 0, length, 1.0 / 4)[0], device=DEV) if mask else \
            torch.full((length,), 1.0 / (4 * length), device=DEV)
        batch = sds.collate(sds.global_rows(sels))
        if fused:
            vl = tr._fused.run(batch, mw, 1.0, normalize_first=False, train=False)
            losses = tr._fused.run(batch, mw, 1.0, normalize_first=True, train=True, opt_step=False)
            # validation form == training form's values when nothing is rescaled in between
            vl2 = tr._fused.run(batch, mw, 1.0, normalize_first=False, train=False)
            for k in ("spectral_loss", "sparsity_loss"):
                assert np.allclose(vl2[k].cpu().numpy(), losses[k].cpu().numpy(), rtol=2e-5), k
            assert not np.allclose(vl["spectral_loss"].cpu().numpy(), vl2["spectral_loss"].cpu().numpy(), rtol=1e-3)
        else:
            tr.optimizer.zero_grad(set_to_none=True)
            losses = tr._step_losses(batch, mask_prenorm=mw, defer_total=True, normalize_first=True)
            heads = losses.pop("_heads")
            torch.autograd.backward(heads, [torch.ones(len(BANDS), device=DEV)] * 2)
            tr.optimizer.pack_grads()
            losses["_total"] = heads[0].detach() + heads[1].detach()
        grad = tr.optimizer.flat_grad.detach().cpu().numpy().copy()
        tr.optimizer.step()
        res[fused] = ({k: v.detach().cpu().numpy() for k, v in losses.items()}, grad,
                      tr.optimizer.flat_param.detach().cpu().numpy().copy(), tr.optimizer)
    for k, v in res[False][0].items():
        assert np.allclose(res[True][0][k], v, rtol=2e-5, atol=1e-7), (k, res[True][0][k], v)
    ga, gb = res[True][1], res[False][1]
    # per leaf (the leaves' gradients differ by orders of magnitude)
    off = 0
    for p in res[False][3]._params:
        sl = slice(off, off + p.numel())
        assert np.abs(ga[sl] - gb[sl]).max() < 2e-4 * np.abs(gb[sl]).max(), (off, np.abs(ga[sl] - gb[sl]).max(), np.abs(gb[sl]).max())
        off += p.numel()
    assert rel_err(res[True][2], res[False][2]) < 1e-4


def _build_net8(q):
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    torch.manual_seed(300 + q)
    base = [173, 179, 181, 191, 193, 197, 199, 211, 223, 227, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283,
            293, 401]
    fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
    of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
    T60 = np.linspace(0.2, 0.5, G)[None, :]
    return DiffGFDNVarReceiverPos(FS, G, [d + 2 * q for d in base], DEV, fl, of, use_absorption_filters=False,
                                  common_decay_times=T60, use_colorless_loss=True).to(DEV)


@pytest.mark.parametrize("mask", [True, False])
def test_fused_bank_step_eight_line_blocks_equals_autograd_bank_step(mask):
    """BASELINE.json configs[4]'s layout (blocks of 8 lines): the explicit step on the 17-polynomial records evaluated on
    the matrix cores (csrc/blocktf8.hip) against the autograd path on the per-bin elimination kernels -- per-band
    losses, every gradient, the post-Adam state."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    sels = [[0, 3, 5, 7], [1, 2, 8, 11], [4, 6, 9, 10]]
    res = {}
    for fused in (True, False):
        BandBankTrainer.use_fused = fused
        try:
            nets = [_build_net8(q) for q in range(len(BANDS))]
            data = [_build_data(q) for q in range(len(BANDS))]
            filt = torch.tensor(_band_filters(), device=DEV).to(torch.complex64)
            bank = BandBank(nets)
            tr = BandBankTrainer(bank, _tc(mask), subband_filter_freq_resp=filt, stft_win=WIN, band_names=BANDS)
        finally:
            BandBankTrainer.use_fused = True
        assert bank.num_delay_lines_per_group == 8 and (tr._fused is not None) == fused
        sds = BandStackedDataset([d for _, d in data])
        start, length = tr._decay_window(NFFT // 2 + 1)
        sds.precompute_decay_targets(WIN, start, length)
        mw = torch.tensor(philox_mask(99, 0, length, 1.0 / 4)[0], device=DEV) if mask else \
            torch.full((length,), 1.0 / (4 * length), device=DEV)
        batch = sds.collate(sds.global_rows(sels))
        if fused:
            losses = tr._fused.run(batch, mw, 1.0, normalize_first=True, train=True, opt_step=False)
        else:
            tr.optimizer.zero_grad(set_to_none=True)
            losses = tr._step_losses(batch, mask_prenorm=mw, defer_total=True, normalize_first=True)
            heads = losses.pop("_heads")
            torch.autograd.backward(heads, [torch.ones(len(BANDS), device=DEV)] * 2)
            tr.optimizer.pack_grads()
            losses["_total"] = heads[0].detach() + heads[1].detach()
        grad = tr.optimizer.flat_grad.detach().cpu().numpy().copy()
        tr.optimizer.step()
        res[fused] = ({k: v.detach().cpu().numpy() for k, v in losses.items()}, grad,
                      tr.optimizer.flat_param.detach().cpu().numpy().copy(), tr.optimizer)
    for k, v in res[False][0].items():
        assert np.allclose(res[True][0][k], v, rtol=5e-5, atol=1e-7), (k, res[True][0][k], v)
    ga, gb = res[True][1], res[False][1]
    off = 0
    for p in res[False][3]._params:
        sl = slice(off, off + p.numel())
        # (two float32 paths against each other; dL/dM is the skew part of the matrix-exponential adjoint, which
        # amplifies the rounding of dL/d(Q Q) on BOTH sides: DESIGN.md section 2)
        within(np.abs(ga[sl] - gb[sl]).max() / np.abs(gb[sl]).max(), 1e-3, ("bank L443", off))
        off += p.numel()
    assert rel_err(res[True][2], res[False][2]) < 1e-4


def test_bank_training_loop_checkpoints_and_band_freeze(tmp_path):
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    nets = [_build_net(q) for q in range(len(BANDS))]
    data = [_build_data(q, R=14) for q in range(len(BANDS))]
    filt = torch.tensor(_band_filters(), device=DEV).to(torch.complex64)
    tc = _tc(True, max_epochs=2)
    tc.train_dir = str(tmp_path)
    bank = BandBank(nets)
    tr = BandBankTrainer(bank, tc, subband_filter_freq_resp=filt, stft_win=WIN, band_names=[int(f) for f in BANDS])
    sds = BandStackedDataset([d for _, d in data])
    g = torch.Generator().manual_seed(0)
    splits = [torch.randperm(14, generator=g).tolist() for _ in BANDS]
    train = [s[:10] for s in splits]           # 2 full batches + a ragged tail of 2 per band
    valid = [s[10:] for s in splits]
    tr.train(sds, train, valid, batch_size=4, log=False)
    assert len(tr.train_loss[0]) == 2 and all(np.isfinite(tr.train_loss[q]).all() for q in range(len(BANDS)))
    for q, f in enumerate(BANDS):
        sd = torch.load(tmp_path / f"band_{int(f)}" / "checkpoints" / "model_e1.pt")
        fresh = _build_net(q)
        fresh.load_state_dict(sd, strict=True)                      # reference-shaped keys
        for k, v in nets[q].state_dict().items():
            assert torch.equal(v.cpu(), sd[k].cpu()), (q, k)
    # a stopped band no longer moves, the others do
    tr.optimizer.band_active[1] = False
    tr.optimizer.sync_lr()
    before = [{k: v.detach().clone() for k, v in n.state_dict().items()} for n in nets]
    b = sds.collate(sds.global_rows([t[:4] for t in train]))
    tr.normalize(b)
    frozen_after_norm = {k: v.detach().clone() for k, v in nets[1].state_dict().items()}
    tr.train_step(b)
    for k, v in nets[1].state_dict().items():
        assert torch.equal(v, frozen_after_norm[k]), k
    assert not torch.equal(nets[0].state_dict()["feedback_loop.M"], before[0]["feedback_loop.M"])


@pytest.mark.parametrize("normalize", [False, True])
def test_fused_colorless_branch_equals_separate_kernels(normalize):
    """gfdn_subfdn_colorless_fwd / spectral_stats_binmajor / subfdn_colorless_bwd against the chain they replace:
    subfdn_normalize, solve (raw M), output stage on eye(G), spectral statistics and the two backward kernels."""
    from diffgfdn_amd import hip_ops as ops
    g = torch.Generator().manual_seed(11)
    nblk, n, K = 6, 4, 3001
    N = nblk * n
    z = torch.tensor(np.exp(1j * 2 * np.pi * np.fft.rfftfreq(2 * (K - 1))), device=DEV)
    turns, _ = ops.zprep(z)
    M = ((2 * torch.rand(nblk, n, n, generator=g) - 1) / 2).to(DEV)
    delays = torch.tensor([173., 181, 191, 193] * nblk, device=DEV) + torch.arange(N, device=DEV)
    b0 = ((2 * torch.randn(N, generator=g) - 1) / N).to(DEV)
    c0 = ((2 * torch.randn(N, generator=g) - 1) / N).to(DEV)
    ones = torch.ones(N, device=DEV)
    eye = torch.eye(nblk, device=DEV)
    # separate chain
    b, c = b0.clone(), c0.clone()
    if normalize:
        ops.subfdn_normalize(turns, None, M, delays, b, c)
    Y = ops.solve_fwd(turns, None, M, delays, ones, b)
    S = ops.compose_fwd(Y, c, eye, n)                                   # (nblk, K)
    _, loss, gS = ops.spectral_stats(S, True, 0.7, True)
    gY, gc, _ = ops.compose_bwd(Y, c, eye, n, gS)
    gM, gb, _ = ops.solve_bwd(turns, None, M, delays, ones, b, gY, Y=Y)
    # fused
    b2, c2 = b0.clone(), c0.clone()
    Y2, S2, en = ops.subfdn_colorless_fwd(turns, None, M, delays, b2, c2, normalize)
    loss2, gS2 = ops.spectral_stats_binmajor(S2, en if normalize else None, True, 0.7, True)
    gM2, gb2, gc2 = ops.subfdn_colorless_bwd(turns, None, M, delays, b2, c2, en if normalize else None, Y2, gS2)
    assert rel_err(b2.cpu(), b.cpu()) < 2e-6 and rel_err(c2.cpu(), c.cpu()) < 2e-6
    assert rel_err(loss2.cpu(), loss.cpu()) < 1e-5
    assert rel_err(torch.view_as_real(gS2.T.contiguous()).cpu(), torch.view_as_real(gS).cpu()) < 1e-5
    assert rel_err(gM2.cpu(), gM.cpu()) < 2e-5
    assert rel_err(gb2.cpu(), gb.cpu()) < 2e-5
    assert rel_err(gc2.cpu(), gc.cpu()) < 2e-5


def test_bank_split_graph_path_for_data_parallel():
    """N > 1 replays two graphs with the all-reduce of the flat gradient buffer (ALL bands) between them;
    with an identity stand-in the result must equal the single-graph path bit for bit."""
    out = {}
    sel = [[[0, 3, 5, 7], [1, 2, 8, 11], [4, 6, 9, 10]], [[1, 2, 4, 6], [0, 5, 7, 9], [3, 8, 10, 11]]]
    for mode in ("single", "split_identity"):
        nets, data, filt, bank, tr, sds, _ = _bank_setup()
        if mode != "single":
            tr._allreduce = lambda: None
        step = tr.graphed(sds, 4, mask_seed=5).capture(sds.global_rows(sel[0]))
        assert (step.graph_b is not None) == (mode != "single")
        vals = [step(sds.global_rows(s))["_total"].cpu().numpy().copy() for s in sel]
        out[mode] = (vals, [{k: v.detach().cpu().clone() for k, v in n.state_dict().items()} for n in nets])
    for a, b in zip(out["single"][0], out["split_identity"][0]):
        assert np.array_equal(a, b)
    for q in range(len(BANDS)):
        for k, v in out["single"][1][q].items():
            assert torch.equal(v, out["split_identity"][1][q][k]), (q, k)


def test_bank_allreduce_captured_inside_the_graph():
    """Data-parallel step as ONE graph: the all-reduce of the gradient bucket (gradients + loss slots) is captured
    with the kernels.  A one-rank RCCL group makes the collective the identity: results must equal the plain
    single-process graph bit for bit, the reported losses must come back from the reduced slots."""
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        import os
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        out = {}
        sel = [[[0, 3, 5, 7], [1, 2, 8, 11], [4, 6, 9, 10]], [[1, 2, 4, 6], [0, 5, 7, 9], [3, 8, 10, 11]]]
        for mode in ("single", "in_graph", "split"):
            nets, data, filt, bank, tr, sds, _ = _bank_setup()
            if mode != "single":
                opt = tr.optimizer
                tr._allreduce = lambda opt=opt: dist.all_reduce(opt.bucket)
                tr.allreduce_in_graph = mode == "in_graph"
            step = tr.graphed(sds, 4, mask_seed=5).capture(sds.global_rows(sel[0]))
            if mode == "in_graph":
                assert step.allreduce_in_graph and step.graph_b is None
                probe_nodes = step.collective_probe_nodes
            if mode == "split":
                assert step.graph_b is not None
            vals = []
            for s in sel:
                r = step(sds.global_rows(s))
                vals.append({k: v.detach().cpu().numpy().copy() for k, v in r.items()})
            out[mode] = (vals, [{k: v.detach().cpu().clone() for k, v in n.state_dict().items()} for n in nets])
        for mode in ("in_graph", "split"):
            for a, b in zip(out["single"][0], out[mode][0]):
                for k in a:
                    assert np.allclose(a[k], b[k], rtol=1e-6, atol=0), (mode, k)      # (total: a different order of adds)
            for q in range(len(BANDS)):
                for k, v in out["single"][1][q].items():
                    assert torch.equal(v, out[mode][1][q][k]), (mode, q, k)
        # What this proves: the step's structure with the collective inside the graph, and that the reported losses come
        # back from the reduced slots.  Whether a COLLECTIVE was captured is a separate question: on a one-rank group RCCL
        # turns the in-place all-reduce into nothing and the probe's graph has no node -- then there is nothing to assert.
        if not probe_nodes:
            pytest.skip(f"the one-rank all-reduce left {probe_nodes} node(s) in the probe graph: values and structure were "
                        "checked above, capture of a real collective needs more than one rank (driver's N > 1 bench)")
        assert probe_nodes >= 1
    finally:
        if created:
            dist.destroy_process_group()


def test_irfft_slot_order_equals_natural_order():
    """n = 65 537: the slot-ordered entry points (no gather / scatter on the spectrum side) against the natural
    ones on the permuted data; the order itself against 3^s mod n."""
    from diffgfdn_amd import hip_ops as ops
    n = 65537
    bins, conj = ops.irfft_slot_order(n, DEV)
    half = (n - 1) // 2
    assert bins.shape == (half,) and int(bins.min()) == 1 and int(bins.max()) == half
    assert sorted(bins.tolist()) == list(range(1, half + 1))                 # a permutation of 1..(n-1)/2
    for s_ in (0, 1, 2, 17, 4099, half - 1):
        k = pow(3, s_, n)
        assert int(bins[s_]) == (k if k <= half else n - k) and bool(conj[s_]) == (k > half)
    assert ops.irfft_slot_order(257, DEV) is None                           # only the 128 x 512 geometry has it
    g = torch.Generator().manual_seed(3)
    B = 3
    X = torch.view_as_complex(torch.randn(B, half + 1, 2, generator=g)).to(DEV)
    Xs = torch.cat([X[:, :1], torch.where(conj, X[:, bins].conj(), X[:, bins])], dim=1).contiguous()
    x = ops.irfft_odd_fwd(X, n)
    xs = ops.irfft_odd_fwd(Xs, n, slots=True)
    assert rel_err(xs.cpu(), x.cpu()) < 1e-6
    gx = torch.randn(B, n, generator=g).to(DEV)
    gx2 = torch.randn(B, n, generator=g).to(DEV)
    gX = ops.irfft_odd_bwd(gx, n, half + 1, gx2)
    gXs = ops.irfft_odd_bwd(gx, n, half + 1, gx2, slots=True)
    want = torch.cat([gX[:, :1], torch.where(conj, gX[:, bins].conj(), gX[:, bins])], dim=1)
    assert rel_err(torch.view_as_real(gXs).cpu(), torch.view_as_real(want).cpu()) < 1e-6


def test_bank_step_slot_order_equals_natural_order_full_size():
    """nfft = 131 072 (K = 65 537, the north-star length): one bank step with the main branch evaluated on the
    irfft's slot-ordered grid against the same step on the natural grid -- losses and every gradient."""
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    from diffgfdn_amd.config import (CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig,
                                     SubbandProcessingConfig, TrainerConfig)
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    from diffgfdn_amd.synthetic import synthetic_room
    from scipy.signal import firwin
    fs, nfft, G, nper, R, B = 32000.0, 131072, 4, 4, 6, 2
    K = nfft // 2 + 1
    centres = (250.0, 1000.0)
    delays = [[641, 701, 809, 907, 1009, 1103, 1201, 1301, 1399, 1409, 1423, 1427, 1429, 1433, 1439, 1601],
              [643, 709, 811, 911, 1013, 1109, 1213, 1303, 1381, 1411, 1423, 1427, 1447, 1451, 1453, 1601]]
    filt = torch.tensor(np.stack([np.fft.rfft(firwin(1025, [f / np.sqrt(2), f * np.sqrt(2)], pass_zero=False,
                                                     fs=fs), n=nfft) for f in centres]), device=DEV).to(torch.complex64)
    res = {}
    for mode in (True, False):
        nets, dss = [], []
        for q in range(2):
            room = synthetic_room(R, G, fs, 40000, seed=40 + q)
            dss.append(MultiRIRDataset(DEV, RoomDataset(G, fs, room["source_position"], room["receiver_position"],
                                                        room["rirs"], room["common_decay_times"], nfft=nfft, device=DEV)))
            torch.manual_seed(200 + q)
            fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
            of = OutputFilterConfig(use_svfs=False, num_hidden_layers=2, num_neurons_per_layer=16, num_fourier_features=4)
            nets.append(DiffGFDNVarReceiverPos(fs, G, delays[q], DEV, fl, of, use_absorption_filters=False,
                                               common_decay_times=room["common_decay_times"],
                                               use_colorless_loss=True).to(DEV))
        tc = TrainerConfig(batch_size=B, num_freq_bins=nfft, lr=1e-3, io_lr=1e-2, use_colorless_loss=True,
                           use_asym_spectral_loss=True, edc_loss_weight=10.0, sparsity_loss_weight=2.0,
                           use_edc_mask=False, train_dir="/tmp/gfdn_bank/t", ir_dir="/tmp/gfdn_bank/a", device="cuda",
                           subband_process_config=SubbandProcessingConfig(centre_frequency=500.0,
                                                                          frequency_range=(63, 8000),
                                                                          num_fraction_octaves=1))
        bank = BandBank(nets)
        tr = BandBankTrainer(bank, tc, subband_filter_freq_resp=filt, band_names=centres)
        tr.use_slot_order = mode
        sds = BandStackedDataset(dss)
        start, length = tr._decay_window(K)
        sds.precompute_decay_targets(4096, start, length)
        batch = sds.collate(sds.global_rows([[0, 3], [1, 4]]))
        tr.normalize(batch)
        tr.optimizer.zero_grad(set_to_none=True)
        losses = tr._step_losses(batch, draw_mask=False, defer_total=True)
        heads = losses.pop("_heads")
        torch.autograd.backward(heads, [torch.ones(2, device=DEV)] * 2)
        tr.optimizer.pack_grads()
        res[mode] = ({k: v.detach().cpu().numpy() for k, v in losses.items()}, tr.optimizer.flat_grad.cpu().numpy().copy())
    for k in res[True][0]:
        assert np.allclose(res[True][0][k], res[False][0][k], rtol=2e-5, atol=0), (k, res[True][0][k], res[False][0][k])
    ga, gb = res[True][1], res[False][1]
    # two float32 pipelines with different summation orders over 65 537 bins: the most cancellation-prone gradient
    # component sits at 1.8e-4 .. 2.4e-4 of the largest one depending on the data (every other entry at 1e-7)
    within(np.abs(ga - gb).max() / np.abs(gb).max(), 5e-4, "bank L675")
    assert np.median(np.abs(ga - gb)) < 1e-6 * np.abs(gb).max()


@pytest.mark.parametrize("B", [4, 3])
def test_pair_interleaved_kernels_equal_per_item_kernels(B):
    """Two items per complex transform / pair-interleaved time signals (irfft pairs, STFT pairs, EDC pairs)
    against the per-item kernels on the same data; odd batch = last pair half empty."""
    from diffgfdn_amd import hip_ops as ops
    n, win, start, L = 65537, 4096, 640, 47360
    half = (n - 1) // 2
    g = torch.Generator().manual_seed(B)
    Xs = torch.view_as_complex(torch.randn(B, half + 1, 2, generator=g)).to(DEV)
    npairs = (B + 1) // 2

    def split(t2):                      # (pairs, n, 2) -> (B, n)
        return t2.permute(0, 2, 1).reshape(2 * npairs, -1)[:B].contiguous()

    def join(t):                        # (B, n) -> (pairs, n, 2), zero partner
        pad = torch.zeros((2 * npairs, t.shape[1]), dtype=t.dtype, device=t.device)
        pad[:B] = t
        return pad.reshape(npairs, 2, -1).permute(0, 2, 1).contiguous()

    x = ops.irfft_odd_fwd(Xs, n, slots=True)
    x2 = ops.irfft_odd_fwd(Xs, n, slots=True, pairs=True)
    assert x2.shape == (npairs, n, 2)
    assert rel_err(split(x2).cpu(), x.cpu()) < 2e-6
    if B % 2:
        assert float(x2[-1, :, 1].abs().max()) < 1e-6 * float(x.abs().max())
    # adjoint with two inputs
    ga = torch.randn(B, n, generator=g).to(DEV)
    gb = torch.randn(B, n, generator=g).to(DEV)
    want = ops.irfft_odd_bwd(ga, n, half + 1, gb, slots=True)
    got = ops.irfft_odd_pairs_bwd(join(ga), n, B, join(gb))
    assert rel_err(torch.view_as_real(got).cpu(), torch.view_as_real(want).cpu()) < 2e-6
    # STFT
    xs = x.contiguous()
    xj = join(xs)
    P = ops.stft_power(xs, win)
    buf = torch.full_like(xj, 7.0)
    Pp = ops.stft_power_pairs(xj, B, win, zero_buf=buf)
    assert rel_err(Pp.cpu(), P.cpu()) < 2e-6 and float(buf.abs().max()) == 0.0
    gP = torch.rand(P.shape, generator=g).to(DEV)
    gx = ops.stft_power_bwd(xs, win, gP, torch.zeros_like(xs))
    g2 = ops.stft_power_pairs_bwd(xj, B, win, gP, out=torch.full_like(xj, 7.0))     # stored: no clearing needed
    assert rel_err(split(g2).cpu(), gx.cpu()) < 2e-6
    if B % 2:
        assert float(g2[-1, :, 1].abs().max()) == 0.0
    base = join(gb)
    keep = base.clone()
    g3 = ops.stft_power_pairs_bwd(xj, B, win, gP, base=base, out=base)               # base + adjoint, in place
    assert g3.data_ptr() == base.data_ptr()
    assert rel_err(split(g3).cpu(), (gx + gb).cpu()) < 2e-6
    g4 = ops.stft_power_pairs_bwd(xj, B, win, gP, base=keep)
    assert torch.equal(g4, g3)
    # the two launches on their own, base added by the second: the same sums in the same order
    g5 = ops.stft_power_pairs_bwd(xj, B, win, gP, out=torch.full_like(xj, 3.0), phase=0)
    g5 = ops.stft_power_pairs_bwd(xj, B, win, gP, base=keep, out=g5, phase=1)
    assert torch.equal(g5, g3)
    # EDC
    tgt = ops.edc_target(torch.randn(B, n, generator=g).to(DEV) * 0.01, start, L)
    mw = (torch.rand(L, generator=g) > 0.5).float().to(DEV) / (B * L / 2)
    li, ge = ops.edc_loss(xs, start, L, tgt, mw, 1.0, 10.0, True)
    lip, gep = ops.edc_loss_pairs(xj, B, start, L, tgt, mw, 1.0, 10.0, True)
    assert rel_err(lip.cpu(), li.cpu()) < 1e-6
    assert rel_err(split(gep).cpu(), ge.cpu()) < 1e-6
    if B % 2:
        assert float(gep[-1, :, 1].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------
# Bands whose longest decay times differ: every band's EDC window ends at ITS T60max (reference trainer.py:56-59,
# run_subband_training_treble.py:286) -- per-item window lengths in the EDC kernels, one row of mask weights per band.
T60MAX = (0.3, 0.4, 0.5)


def _band_rows(seed, step, lens, gb):
    """(bands, Lmax) pre-normalised weights of gfdn_draw_mask_banded, restated: ONE draw of max(lens) bits, band q
    keeps the first lens[q] of them and divides by (gb x its own count); plus the kept indices per band."""
    Lmax = max(lens)
    bits = (philox_mask(seed, step, Lmax, 1.0)[0] > 0)
    rows = np.zeros((len(lens), Lmax), dtype=np.float32)
    keeps = []
    for q, L in enumerate(lens):
        b = bits[:L]
        rows[q, :L] = b.astype(np.float32) * (np.float32(1.0 / gb) / np.float32(b.sum()))
        keeps.append(torch.argwhere(torch.tensor(b)))
    return rows, keeps


@pytest.mark.parametrize("fused", [True, False])
def test_bank_with_distinct_decay_windows_equals_band_steps_and_oracle(fused):
    """Three bands with T60max = 0.3 / 0.4 / 0.5 s: one bank step (explicit launch sequence and the autograd fallback)
    against every band's OWN VarReceiverPosTrainer step -- whose EDC window is the band's own -- and against the CPU
    oracle of the reference step, on the same per-band masks."""
    from diffgfdn_amd.bandbank import BandBankTrainer
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    from oracle import gfdn_oracle as orc
    from oracle.cpu_trainer import OracleGridTrainer
    Kf = NFFT // 2 + 1
    BandBankTrainer.use_fused = fused
    try:
        nets, data, filt, bank, tr, sds, (start, length) = _bank_setup(t60max=T60MAX)
    finally:
        BandBankTrainer.use_fused = True
    assert (tr._fused is not None) == fused
    lens = tr._band_windows(Kf)
    assert lens == [int(t * FS) - start for t in T60MAX] and length == max(lens)
    assert sds.edc_store[0] == (start, tuple(lens)) and sds.edc_store[1].shape[1] == length
    sels = [[0, 3, 5, 7], [1, 2, 8, 11], [4, 6, 9, 10]]
    rows_np, keeps = _band_rows(99, 0, lens, 4)
    mw = torch.tensor(rows_np, device=DEV)
    sd0 = [{k: v.detach().cpu().clone() for k, v in net.state_dict().items()} for net in nets]
    batch = sds.collate(sds.global_rows(sels))
    if fused:
        losses = tr._fused.run(batch, mw, 1.0, normalize_first=True, train=True)
    else:
        tr.normalize(batch)
        tr.optimizer.zero_grad(set_to_none=True)
        losses = tr._step_losses(batch, mask_prenorm=mw, defer_total=True)
        heads = losses.pop("_heads")
        torch.autograd.backward(heads, [torch.ones(len(BANDS), device=DEV)] * 2)
        tr.optimizer.pack_grads()
        tr.optimizer.step()
    torch.cuda.synchronize()
    got = {k: v.detach().cpu().numpy() for k, v in losses.items() if k.endswith("_loss")}

    for q in range(len(BANDS)):
        ref_net = _build_net(q, T60MAX[q])
        ref_net.load_state_dict(sd0[q], strict=True)
        rtr = VarReceiverPosTrainer(ref_net, _tc(True), subband_filter_freq_resp=filt[q], stft_win=WIN, capturable=True)
        assert rtr._decay_window(Kf) == (start, lens[q])
        _, ds = _build_data(q, t60max=T60MAX[q])
        ds.precompute_decay_targets(WIN, start, lens[q])
        b = ds.collate(sels[q], lean=True)
        rtr.normalize(b)
        rtr.optimizer.zero_grad(set_to_none=True)
        rl = rtr._step_losses(b, mask_prenorm=mw[q, :lens[q]].contiguous())
        rl.pop("_total").backward()
        rtr.optimizer.step()
        for k, v in rl.items():
            assert abs(float(v) - got[k][q]) <= (1e-5 if fused else 1e-6) * abs(float(v)) + 1e-9, (q, k, float(v), got[k][q])
        for k, v in ref_net.state_dict().items():
            assert rel_err(nets[q].state_dict()[k].detach().cpu(), v.detach().cpu()) < (5e-4 if fused else 1e-5), (q, k)

    for q in range(len(BANDS)):
        room, dq = data[q]
        sd = sd0[q]
        lin, norm = [], []
        for i in range(64):
            k = f"output_scalars.mlp.model.{i}.weight"
            if k in sd:
                (lin if sd[k].ndim == 2 else norm).append((sd[k].clone(), sd[f"output_scalars.mlp.model.{i}.bias"].clone()))
        p = orc.GridModelParams(FS, _delays(q), G, sd["input_gains"].clone(), sd["output_gains"].clone(),
                                sd["feedback_loop.M"].clone(), sd["feedback_loop.alpha"].clone(),
                                np.linspace(0.2, T60MAX[q], G)[None, :], lin, norm, 4)
        otr = OracleGridTrainer(p, lr=1e-3, io_lr=1e-2, edr_weight=1.0, edc_weight=10.0, spectral_weight=1.0,
                                sparsity_weight=2.0, use_asym=True, win=WIN, hop=WIN // 2,
                                subband_filter=filt[q].cpu().to(torch.complex128))
        idx = torch.tensor(sels[q])
        ob = {"z_values": dq.z_values.cpu(),
              "norm_listener_position": dq.norm_listener_position[idx].cpu(),
              "listener_position": dq.listener_positions[idx].cpu(),
              "target_early_response": dq.early_response_c128(idx).cpu(),
              "target_rir_response": dq.rir_mag_response[idx].cpu().to(torch.complex128)}
        otr.normalize(ob)
        _, oparts = otr.train_step(ob, keeps[q])
        for k, v in oparts.items():
            assert abs(got[k][q] - v) < 1e-4 * abs(v) + 1e-7, (q, k, got[k][q], v)
        for name in ("input_gains", "output_gains"):
            a = getattr(nets[q], name).detach().cpu()
            assert rel_err(a, getattr(p, name).detach()) < 1e-4, (q, name)


def test_graphed_bank_step_with_distinct_decay_windows():
    """The captured step of a bank with three different EDC windows: the mask rows are drawn on the device per band
    (gfdn_draw_mask_banded) inside the graph; replays against the explicit step launched from the host with the restated
    rows: bit-equal gradients and losses."""
    nets, data, filt, bank, tr, sds, (start, length) = _bank_setup(t60max=T60MAX)
    lens = tr._band_windows(NFFT // 2 + 1)
    sel_steps = [[[0, 3, 5, 7], [1, 2, 8, 11], [4, 6, 9, 10]], [[1, 2, 4, 6], [0, 5, 7, 9], [3, 8, 10, 11]]]
    step = tr.graphed(sds, 4, mask_seed=4711).capture(sds.global_rows(sel_steps[0]))
    assert step.maskw.shape == (len(BANDS), length)
    got, got_grad, got_mask = [], [], []
    for s in sel_steps:
        got.append({k: v.detach().cpu().numpy().copy() for k, v in step(sds.global_rows(s)).items()})
        got_grad.append(tr.optimizer.flat_grad.detach().cpu().numpy().copy())
        got_mask.append(step.maskw.detach().cpu().numpy().copy())
    nets2, data2, filt2, bank2, tr2, sds2, _ = _bank_setup(t60max=T60MAX)
    for i, s in enumerate(sel_steps):
        rows_np, _ = _band_rows(4711, i, lens, 4)
        assert np.array_equal(got_mask[i], rows_np)              # integer work + one float division: bit-exact
        out = tr2._fused.run(sds2.collate(sds2.global_rows(s)), torch.tensor(rows_np, device=DEV), 1.0,
                             normalize_first=True, train=True)
        torch.cuda.synchronize()
        assert np.array_equal(tr2.optimizer.flat_grad.cpu().numpy(), got_grad[i]), i
        for k, v in out.items():
            assert np.array_equal(v.detach().cpu().numpy(), got[i][k]), (i, k)


@pytest.mark.parametrize("t60max", [None, T60MAX])
def test_time_domain_output_stage_equals_spectral_output_stage(t60max, float32_direct_store):
    """The explicit step with the output stage in the time domain (csrc/linear.hip: G transforms per band, receivers'
    signals combined from the transformed group responses and the dataset's transformed direct paths) against the same
    step with the output stage formed per receiver in the frequency domain and one transform per receiver: the maps
    commute, so losses and every gradient agree to float32 rounding."""
    from diffgfdn_amd.bankstep import FusedBankStep
    sels = [[0, 3, 5, 7], [1, 2, 8, 11], [4, 6, 9, 10]]
    res = {}
    for lin in (True, False):
        nets, data, filt, bank, tr, sds, (start, length) = _bank_setup(t60max=t60max)
        tr._fused.linear_transforms = lin
        Kf = NFFT // 2 + 1
        lens = tr._band_windows(Kf)
        if lens is None:
            mw = torch.tensor(philox_mask(5, 0, length, 1.0 / 4)[0], device=DEV)
        else:
            mw = torch.tensor(_band_rows(5, 0, lens, 4)[0], device=DEV)
        out = tr._fused.run(sds.collate(sds.global_rows(sels)), mw, 1.0, normalize_first=True, train=True, opt_step=False)
        torch.cuda.synchronize()
        res[lin] = ({k: v.detach().cpu().numpy().copy() for k, v in out.items()},
                    {id_: g.copy() for id_, g in zip(("c", "b", "M", "w"), _grad_views(tr))})
    assert FusedBankStep.linear_transforms
    for k, v in res[False][0].items():
        assert np.allclose(res[True][0][k], v, rtol=2e-6, atol=0), (k, res[True][0][k], v)
    for k, g in res[False][1].items():
        assert np.abs(res[True][1][k] - g).max() <= 2e-5 * np.abs(g).max(), (k, np.abs(res[True][1][k] - g).max(), np.abs(g).max())


def _grad_views(tr):
    flat = tr.optimizer.flat_grad.detach().cpu().numpy()
    out, off = [], 0
    for p in tr.optimizer._params:
        out.append(flat[off:off + p.numel()])
        off += p.numel()
    return out
