"""Round-6 pieces on the MI355X: the float64 transforms for dataset constants (csrc/fft64.hip) against numpy's complex128
transforms, and the band bank's direct-path store built with them against a float64 evaluation of the reference's formula."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from diffgfdn_amd import hip_ops
    return hip_ops


@pytest.mark.parametrize("n,batch", [(65537, 5), (1025, 40), (37, 3), (20011, 33)])
def test_float64_inverse_transform_of_odd_length(ops, n, batch):
    """gfdn_irfft_odd_f64 (Bluestein on radix-2 double passes) == numpy.fft.irfft(X, n) on complex128 spectra to 1e-13 of the
    largest sample -- with and without the filter factor, float64 and float32 output, more rows than one chunk"""
    rng = np.random.default_rng(n)
    h = (n + 1) // 2
    X = rng.standard_normal((batch, h + 3)) + 1j * rng.standard_normal((batch, h + 3))      # (a pitch beyond the bins read)
    f = rng.standard_normal(h) + 1j * rng.standard_normal(h)
    Xd = torch.tensor(X, device=DEV)
    for filt in (None, f):
        ref = np.fft.irfft(X[:, :h] * (1.0 if filt is None else filt[None, :]), n=n, axis=-1)
        fd = None if filt is None else torch.tensor(filt, device=DEV)
        got = ops.irfft_odd_f64(Xd, n, filt=fd, out_dtype=torch.float64).cpu().numpy()
        assert np.abs(got - ref).max() < 1e-13 * np.abs(ref).max()
        got32 = ops.irfft_odd_f64(Xd, n, filt=fd).cpu().numpy()
        assert got32.dtype == np.float32 and np.array_equal(got32, ref.astype(np.float32)) or \
            np.abs(got32 - ref.astype(np.float32)).max() <= np.spacing(np.float32(np.abs(ref).max()))


@pytest.mark.parametrize("nfft,T,batch", [(131072, 800, 6), (2048, 2048, 3), (1024, 37, 2)])
def test_float64_forward_transform(ops, nfft, T, batch):
    """gfdn_rfft_pow2_f64 == numpy.fft.rfft(x, nfft) (zero-padded rows, the first kout bins) to 1e-13"""
    rng = np.random.default_rng(nfft + T)
    x = rng.standard_normal((batch, T))
    ref = np.fft.rfft(x, n=nfft, axis=-1)
    got = ops.rfft_pow2_f64(torch.tensor(x, device=DEV), nfft).cpu().numpy()
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-13 * np.abs(ref).max()
    kout = nfft // 4 + 1
    got = ops.rfft_pow2_f64(torch.tensor(x, device=DEV), nfft, kout).cpu().numpy()
    assert np.abs(got - ref[:, :kout]).max() < 1e-13 * np.abs(ref).max()


def test_direct_path_store_is_float64_accurate(ops):
    """BandStackedDataset.direct_time at the north-star size: xd = irfft(rfft(early RIR, nfft)[: K] filt, n = K) (reference
    dataloader.py:250, model.py:618-619, trainer.py:459, losses.py:442-445) against numpy in float64 -- the float64 build
    is the float32 rounding of the exact rows (<= 1 ulp of the row's largest sample), the float32 build of rounds 4-5 is
    3e-7 of it away; on the last tenth of the EDC window that is the difference between 1e-7 and percent-level error."""
    from diffgfdn_amd.bandbank import BandStackedDataset
    from diffgfdn_amd.dataloader import MultiRIRDataset, RoomDataset
    rng = np.random.default_rng(3)
    fs, nfft, R, T = 32000, 131072, 6, 64000
    K = nfft // 2 + 1
    t = np.arange(T) / fs
    rooms = []
    for q in range(2):
        rirs = rng.standard_normal((R, T)) * np.exp(-6.908 * t / (0.4 + 0.5 * q))[None, :]
        room = RoomDataset(2, fs, np.zeros((1, 3)), rng.uniform(0, 5, (R, 3)), rirs.copy(), np.array([0.4, 0.9]),
                           band_centre_hz=500.0 * (q + 1), nfft=nfft, device=DEV)
        rooms.append((rirs, room))
    sds = BandStackedDataset([MultiRIRDataset(DEV, room) for _, room in rooms])
    filt = torch.tensor(rng.standard_normal((2, K)) + 1j * rng.standard_normal((2, K)), device=DEV).to(torch.complex64)
    xd = sds.direct_time(filt, K).cpu().numpy()
    BandStackedDataset.direct_time_f64 = False
    try:
        xd32 = sds.direct_time(filt, K).cpu().numpy()
    finally:
        BandStackedDataset.direct_time_f64 = True
    f128 = filt.cpu().numpy().astype(np.complex128)
    h = (K + 1) // 2
    for q, (_, room) in enumerate(rooms):
        early = room.early_rir_time.cpu().numpy()
        ref = np.fft.irfft(np.fft.rfft(early, n=nfft, axis=-1)[:, :h] * f128[q, None, :h], n=K, axis=-1)
        top = np.abs(ref).max(axis=-1, keepdims=True)
        rows = slice(q * R, (q + 1) * R)
        e64 = np.abs(xd[rows] - ref) / top
        e32 = np.abs(xd32[rows] - ref) / top
        assert e64.max() < 1.2e-7, e64.max()                   # (float32 rounding of the exact value)
        assert e32.max() > 5 * e64.max()                       # (what the float32 transforms leave)
        # the accessor the full-size oracle tests feed the oracle from
        ds = sds.datasets[q]
        E = ds.early_response_c128(torch.arange(R)).cpu().numpy()
        assert np.abs(E - np.fft.rfft(early, n=nfft, axis=-1)).max() < 1e-13 * np.abs(E).max()


# ---------------------------------------------------------------------------------------------
# bands with their own gain networks (reference run_subband_training_treble.py:61-73)
MIXED = [(8, 1), (16, 1), (16, 5), (128, 3)]          # (neurons, hidden layers) per band


@pytest.mark.parametrize("Bper", [8, 5])
def test_gain_networks_of_different_sizes_in_one_launch(ops, Bper):
    """gfdn_mlp_gains_bands_fwd / _bwd (a table of per-band layer sizes, one launch) == the per-band launches, bit for
    bit: bands on the wave-per-receiver form (<= 64 neurons, whole groups of 8 receivers) and on the workgroup-per-receiver
    form (128 neurons: parameters read from memory) in the same launch; summed gradients and partial rows with column
    scales; float64 torch autograd of the same stack for the 128-neuron band."""
    g = torch.Generator(device="cpu").manual_seed(7 + Bper)
    nb, R, G, F = len(MIXED), 11, 3, 20
    H, nh = [m[0] for m in MIXED], [m[1] for m in MIXED]
    counts = ops.mlp_bands_param_counts(H, nh, F, G)
    w = torch.cat([0.3 * torch.randn(c, generator=g) / np.sqrt(h) for c, h in zip(counts, H)]).to(DEV)
    pos = torch.rand(nb * R, 3, generator=g, dtype=torch.float64).to(DEV)
    fpi = (torch.exp(torch.linspace(0, np.log(32.0), F)) * np.pi).to(torch.float32).to(DEV)
    rows = torch.tensor([q * R + int(i) for q in range(nb) for i in torch.randperm(R, generator=g)[:Bper]], device=DEV)
    gains, xhat, rstd = ops.mlp_gains_fwd(pos, fpi, w, H, nh, G, -1.0, 1.0, rows, nbands=nb)
    gg = torch.randn(nb * Bper, G, generator=g).to(DEV)
    parts = torch.randn(nb * Bper * G, 7, generator=g).to(DEV)
    cs = (torch.rand(nb * G, generator=g) + 0.5).to(DEV)
    gw = ops.mlp_gains_bwd(pos, fpi, w, H, nh, G, -1.0, 1.0, gains, xhat, rstd, gg, rows, nbands=nb)
    gwp = ops.mlp_gains_bwd(pos, fpi, w, H, nh, G, -1.0, 1.0, gains, xhat, rstd, None, rows, nbands=nb,
                            ggains_parts=parts, colscale=cs)
    wo = xo = ro = 0
    for q in range(nb):
        sl = slice(q * Bper, (q + 1) * Bper)
        rq = rows[sl].contiguous()
        nl = 1 + nh[q]
        wq = w[wo:wo + counts[q]].contiguous()
        gq, xq, sq = ops.mlp_gains_fwd(pos, fpi, wq, H[q], nh[q], G, -1.0, 1.0, rq)
        assert torch.equal(gains[sl], gq), q
        assert torch.equal(xhat[xo:xo + Bper * nl * H[q]], xq.reshape(-1)) and torch.equal(rstd[ro:ro + Bper * nl], sq.reshape(-1))
        gwq = ops.mlp_gains_bwd(pos, fpi, wq, H[q], nh[q], G, -1.0, 1.0, gq, xq, sq, gg[sl].contiguous(), rq)
        if H[q] <= 64:
            assert torch.equal(gw[wo:wo + counts[q]], gwq), q
        else:       # (512 threads split a layer's neurons into four groups, the band's own launch into fewer: another order)
            assert float((gw[wo:wo + counts[q]] - gwq).abs().max()) <= 1e-5 * float(gwq.abs().max()), q
        # partial rows x column scale == the summed, scaled rows (another summation order: float32 rounding)
        ggq = (parts.view(nb * Bper, G, -1).sum(-1)[sl] * cs[q * G:(q + 1) * G][None, :]).contiguous()
        gwq2 = ops.mlp_gains_bwd(pos, fpi, wq, H[q], nh[q], G, -1.0, 1.0, gq, xq, sq, ggq, rq)
        assert float((gwp[wo:wo + counts[q]] - gwq2).abs().max()) <= 2e-5 * float(gwq2.abs().max()), q
        wo, xo, ro = wo + counts[q], xo + Bper * nl * H[q], ro + Bper * nl
    # the 128-neuron band against float64 autograd (dnn.py:331-400, gain_filters.py:497-524)
    q = 3
    wq = w[sum(counts[:q]):].double().cpu().requires_grad_()
    x = pos[rows[q * Bper:]].cpu()
    enc = []
    for k in range(F):
        a = float(fpi[k].cpu()) * x
        enc += [torch.sin(a), torch.cos(a)]
    a = torch.cat(enc, dim=1).float().double()
    off, n_in = 0, 6 * F
    for l in range(1 + nh[q]):
        W = wq[off:off + H[q] * n_in].view(H[q], n_in)
        off += H[q] * n_in
        bias, gamma, beta = wq[off:off + H[q]], wq[off + H[q]:off + 2 * H[q]], wq[off + 2 * H[q]:off + 3 * H[q]]
        off += 3 * H[q]
        a = torch.relu(torch.nn.functional.layer_norm(a @ W.T + bias, (H[q],), gamma, beta, 1e-5))
        n_in = H[q]
    out = -1.0 + 2.0 * torch.sigmoid(a @ wq[off:off + G * H[q]].view(G, H[q]).T + wq[off + G * H[q]:])
    assert float((gains[q * Bper:].cpu().double() - out).abs().max()) < 2e-5
    out.backward(gg[q * Bper:].cpu().double())
    ref = wq.grad
    assert float((gw[sum(counts[:q]):].cpu().double() - ref).abs().max()) < 2e-4 * float(ref.abs().max())


def test_bank_with_the_reference_recipes_gain_networks():
    """A bank whose bands have the reference's per-band network sizes (1 x 8, 1 x 16, 5 x 16, 3 x 128;
    run_subband_training_treble.py:61-73): the explicit step (one launch per stage for all bands, the scale inside the
    gains) == the autograd bank step on the per-bin elimination kernels == every band's own trainer step; the bands'
    reference-shaped state dicts stay views of the bank's packed vector."""
    from tests import test_gpu_bank as tb
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    from diffgfdn_amd.config import CouplingMatrixType, FeedbackLoopConfig, OutputFilterConfig
    from diffgfdn_amd.model import DiffGFDNVarReceiverPos
    from diffgfdn_amd.trainer import VarReceiverPosTrainer
    from tests.helpers import philox_mask, rel_err
    nbands = 3
    sizes = [(8, 1), (16, 5), (128, 3)]

    def build(q):
        torch.manual_seed(100 + q)
        fl = FeedbackLoopConfig(coupling_matrix_type=CouplingMatrixType.SCALAR, use_zero_coupling=True)
        of = OutputFilterConfig(use_svfs=False, num_hidden_layers=sizes[q][1], num_neurons_per_layer=sizes[q][0],
                                num_fourier_features=4)
        T60 = np.linspace(0.2, 0.5, tb.G)[None, :]
        return DiffGFDNVarReceiverPos(tb.FS, tb.G, tb._delays(q), DEV, fl, of, use_absorption_filters=False,
                                      common_decay_times=T60, use_colorless_loss=True).to(DEV)

    sels = [[0, 3, 5, 7, 1, 2, 8, 11], [1, 2, 8, 11, 4, 6, 9, 10], [4, 6, 9, 10, 0, 3, 5, 7]]
    filt = torch.tensor(tb._band_filters(), device=DEV).to(torch.complex64)
    res = {}
    for fused in (True, False):
        BandBankTrainer.use_fused = fused
        try:
            nets = [build(q) for q in range(nbands)]
            data = [tb._build_data(q) for q in range(nbands)]
            bank = BandBank(nets)
            assert bank.mixed_networks and bank.output_scalars_w.dim() == 1
            tr = BandBankTrainer(bank, tb._tc(True, ), subband_filter_freq_resp=filt, stft_win=tb.WIN, band_names=tb.BANDS)
        finally:
            BandBankTrainer.use_fused = True
        sds = BandStackedDataset([d for _, d in data])
        start, length = tr._decay_window(tb.NFFT // 2 + 1)
        sds.precompute_decay_targets(tb.WIN, *tr._target_window(tb.NFFT // 2 + 1))
        mw = torch.tensor(philox_mask(99, 0, length, 1.0 / 8)[0], device=DEV)
        sd0 = [{k: v.detach().cpu().clone() for k, v in net.state_dict().items()} for net in nets]
        batch = sds.collate(sds.global_rows(sels))
        if fused:
            losses = tr._fused.run(batch, mw, 1.0, normalize_first=True, train=True, opt_step=False)
        else:
            tr.optimizer.zero_grad(set_to_none=True)
            losses = tr._step_losses(batch, mask_prenorm=mw, defer_total=True, normalize_first=True)
            heads = losses.pop("_heads")
            torch.autograd.backward(heads, [torch.ones(nbands, device=DEV)] * 2)
            tr.optimizer.pack_grads()
            losses["_total"] = heads[0].detach() + heads[1].detach()
        grad = tr.optimizer.flat_grad.detach().cpu().numpy().copy()
        tr.optimizer.step()
        res[fused] = ({k: v.detach().cpu().numpy() for k, v in losses.items()}, grad, tr.optimizer, nets, sd0, data,
                      (start, length), mw)
    for k, v in res[False][0].items():
        assert np.allclose(res[True][0][k], v, rtol=2e-5, atol=1e-7), (k, res[True][0][k], v)
    ga, gb = res[True][1], res[False][1]
    off = 0
    for p in res[False][2]._params:
        sl = slice(off, off + p.numel())
        assert np.abs(ga[sl] - gb[sl]).max() < 2e-4 * np.abs(gb[sl]).max(), (off, np.abs(ga[sl] - gb[sl]).max())
        off += p.numel()
    # every band's own trainer step from the same start
    _, _, _, nets, sd0, data, (start, length), mw = res[True]
    for q in range(nbands):
        ref_net = build(q)
        ref_net.load_state_dict(sd0[q], strict=True)
        rtr = VarReceiverPosTrainer(ref_net, tb._tc(True), subband_filter_freq_resp=filt[q], stft_win=tb.WIN, capturable=True)
        ds = data[q][1]
        ds.precompute_decay_targets(tb.WIN, start, length)
        b = ds.collate(sels[q], lean=True)
        rtr.normalize(b)
        rtr.optimizer.zero_grad(set_to_none=True)
        rl = rtr._step_losses(b, mask_prenorm=mw)
        rl.pop("_total").backward()
        rtr.optimizer.step()
        for k, v in rl.items():
            assert abs(float(v) - res[True][0][k][q]) <= 2e-5 * abs(float(v)) + 1e-9, (q, k)
        for k, v in ref_net.state_dict().items():
            assert rel_err(nets[q].state_dict()[k].detach().cpu(), v.detach().cpu()) < 2e-4, (q, k)


def test_bank_step_above_the_linear_kernels_receiver_limits():
    """66 receivers per band: beyond what the EDR launch on composed spectra takes (hip_ops.spec_supported: <= 64), inside
    what the time-domain output stage takes (lin_supported: B G <= 256) -- the explicit step falls through to the chain on
    the receivers' own signals (ADVICE r4 / VERDICT r5 Missing 6: the predicates existed, no test crossed them).  Losses,
    every gradient and the post-Adam state against the autograd bank step on the per-bin elimination kernels."""
    from tests import test_gpu_bank as tb
    from diffgfdn_amd import hip_ops as ops
    from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset
    from tests.helpers import philox_mask, rel_err
    B, R, nbands = 66, 70, 2
    assert ops.lin_supported(B, tb.G) and not ops.spec_supported(B, tb.G)
    filt = torch.tensor(tb._band_filters()[:nbands], device=DEV).to(torch.complex64)
    rng = np.random.RandomState(3)
    sels = [rng.permutation(R)[:B].tolist() for _ in range(nbands)]
    res = {}
    for fused in (True, False):
        BandBankTrainer.use_fused = fused
        try:
            nets = [tb._build_net(q) for q in range(nbands)]
            data = [tb._build_data(q, R=R) for q in range(nbands)]
            bank = BandBank(nets)
            tr = BandBankTrainer(bank, tb._tc(True, ), subband_filter_freq_resp=filt, stft_win=tb.WIN,
                                 band_names=tb.BANDS[:nbands])
        finally:
            BandBankTrainer.use_fused = True
        assert (tr._fused is not None) == fused
        sds = BandStackedDataset([d for _, d in data])
        start, length = tr._decay_window(tb.NFFT // 2 + 1)
        sds.precompute_decay_targets(tb.WIN, *tr._target_window(tb.NFFT // 2 + 1))
        mw = torch.tensor(philox_mask(99, 0, length, 1.0 / B)[0], device=DEV)
        batch = sds.collate(sds.global_rows(sels))
        if fused:
            losses = tr._fused.run(batch, mw, 1.0, normalize_first=True, train=True, opt_step=False)
        else:
            tr.optimizer.zero_grad(set_to_none=True)
            losses = tr._step_losses(batch, mask_prenorm=mw, defer_total=True, normalize_first=True)
            heads = losses.pop("_heads")
            torch.autograd.backward(heads, [torch.ones(nbands, device=DEV)] * 2)
            tr.optimizer.pack_grads()
            losses["_total"] = heads[0].detach() + heads[1].detach()
        grad = tr.optimizer.flat_grad.detach().cpu().numpy().copy()
        tr.optimizer.step()
        res[fused] = ({k: v.detach().cpu().numpy() for k, v in losses.items()}, grad,
                      tr.optimizer.flat_param.detach().cpu().numpy().copy(), tr.optimizer)
    for k, v in res[False][0].items():
        assert np.allclose(res[True][0][k], v, rtol=2e-5, atol=1e-7), (k, res[True][0][k], v)
    ga, gb = res[True][1], res[False][1]
    off = 0
    for p in res[False][3]._params:
        sl = slice(off, off + p.numel())
        assert np.abs(ga[sl] - gb[sl]).max() < 2e-4 * np.abs(gb[sl]).max(), (off, np.abs(ga[sl] - gb[sl]).max())
        off += p.numel()
    assert rel_err(res[True][2], res[False][2]) < 1e-4


@pytest.mark.parametrize("G,B", [(2, 70), (4, 67)])
def test_edr_one_launch_with_runs_longer_than_a_wave(ops, G, B):
    """gfdn_edr_lin_loss_gsum with more than 64 receivers in one run of a band (k_edr_lin_wave keeps gscale / sum_abs of a run's
    receivers a lane each and refills the table every 64 receivers; an odd run length ends in the loop's tail copy): loss partials,
    the EDR part of dL/drgain and the gradient spectra against a float64 evaluation of losses.py:430-495, and one run against two
    (the same bits per receiver)."""
    gen = torch.Generator(device="cpu").manual_seed(10 * G + B)
    nb, R, nfr, nf = 2, B + 2, 32, 2049
    items = nb * B
    Sd = torch.view_as_complex(torch.randn(nb * R, nfr, nf, 2, generator=gen).to(DEV))
    Stau = torch.view_as_complex(torch.randn(nb * G, nfr, nf, 2, generator=gen).to(DEV))
    rgain = torch.randn(items, G, generator=gen).to(DEV)
    rows = torch.tensor([q * R + int(i) for q in range(nb) for i in torch.randperm(R, generator=gen)[:B]], device=DEV)
    Pt = torch.rand(nb * R, nfr, nf, generator=gen).to(DEV) * 3 + 0.1
    T_db, sum_abs = ops.edr_target(Pt.clone())
    nparts = ops.edr_lin_parts(nf, fused=True)
    parts1 = torch.zeros(items * G, nparts, device=DEV)
    part1, Gs1 = ops.edr_lin_loss_gsum(Sd, rows, Stau, rgain, nb, T_db, sum_abs, 1.5, dots=parts1, col0=0, nsplit=1)
    parts2 = torch.zeros_like(parts1)
    part2, Gs2 = ops.edr_lin_loss_gsum(Sd, rows, Stau, rgain, nb, T_db, sum_abs, 1.5, dots=parts2, col0=0, nsplit=2)
    assert torch.equal(part1, part2) and torch.equal(parts1, parts2)
    # float64: S = Sd[row] + sum_g gain Stau_g; EDR_m = sum_{m' >= m} |S_m'|^2; loss = sum |T - 10 log10(EDR + eps)| / sum |T|
    band = torch.arange(items, device=DEV) // B
    S = (Sd[rows].to(torch.complex128)
         + (rgain.to(torch.complex128)[:, :, None, None] * Stau.view(nb, G, nfr, nf)[band].to(torch.complex128)).sum(1))
    S = S.detach().requires_grad_(True)
    P = S.real ** 2 + S.imag ** 2
    edr = torch.flip(torch.cumsum(torch.flip(P, dims=[1]), dim=1), dims=[1])
    db = torch.clamp(10.0 * torch.log10(edr.abs() + float(np.finfo(np.float32).eps)), min=-200.0)
    li = (T_db[rows].double() - db).abs().sum(dim=(1, 2)) / sum_abs[rows].double()
    (1.5 * li.sum()).backward()
    dS = S.grad                                                                   # dL/dRe + i dL/dIm
    assert float(((part1.sum(1) / sum_abs[rows]).double() - li.detach()).abs().max() / li.detach().abs().max()) < 2e-6
    dots_ref = (Stau.view(nb, G, nfr, nf)[band].to(torch.complex128).conj() * dS[:, None]).real.sum(dim=(2, 3))
    assert float((parts1.sum(1).view(items, G).double() - dots_ref).abs().max() / dots_ref.abs().max()) < 2e-5
    # (L1: where a target sits within float32 rounding of the EDR the sign of |.| differs between float32 and float64, and that
    # cell's column of the gradient with it -- a handful of 8.8 M cells)
    Gs_ref = (rgain.double().view(nb, B, G, 1, 1) * dS.view(nb, B, 1, nfr, nf)).sum(1).reshape(nb * G, nfr, nf)
    for Gs in (Gs1.sum(0), Gs2.sum(0)):
        assert float((Gs.to(torch.complex128) - Gs_ref).abs().sum() / Gs_ref.abs().sum()) < 1e-5
    assert float((Gs1.sum(0) - Gs2.sum(0)).abs().max() / Gs2.sum(0).abs().max()) < 1e-6
