"""The fused decay-loss kernel (csrc/decay.hip: STFT -> EDR term and the whole EDC term, one workgroup per item) against
the float64 maths of the oracle (losses.py:430-495, :201-238) and against the unfused kernels it replaces."""
import numpy as np
import pytest
import torch

from oracle import gfdn_oracle as orc
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from diffgfdn_amd import hip_ops
    return hip_ops


def _signals(items, T, seed, tau):
    rng = np.random.RandomState(seed)
    t = np.arange(T)
    return torch.tensor(rng.randn(items, T) * np.exp(-t / tau) * rng.uniform(0.5, 1.5, (items, 1)))


def _pairs(x):
    """(items, T) float64 -> pair-interleaved (ceil(items / 2), T, 2) float32 on the device"""
    items, T = x.shape
    x2 = torch.zeros((items + 1) // 2 * 2, T, dtype=torch.float32)
    x2[:items] = x.float()
    return x2.view(-1, 2, T).transpose(1, 2).contiguous().to(DEV)


def _edr_db(x, win):
    S = orc.stft_onesided(x, win, win // 2)                               # (items, F, frames)
    P = (S.abs() ** 2).transpose(1, 2)                                    # (items, frames, F)
    E = torch.flip(torch.cumsum(torch.flip(P, dims=[1]), dim=1), dims=[1])
    return P, orc.db(E, is_squared=True)


@pytest.mark.parametrize("items,T,start,length,masked,rows_map", [
    (6, 65537, 640, 47360, True, True),       # the north-star shape: 32 frames, window ends inside the frames
    (5, 65537, 640, 64897, False, False),     # odd batch; the window runs to the LAST sample (beyond the last round's tile)
    (3, 20000, 100, 15000, True, False),      # 9 frames: a partial first round, one-frame group
    (2, 9000, 0, 9000, False, True),          # 4 frames, the window is the whole signal
])
def test_decay_items_fwd_vs_float64(ops, items, T, start, length, masked, rows_map):
    win = 4096
    nrows = items + 3 if rows_map else items
    rows = torch.tensor(np.random.RandomState(1).permutation(nrows)[:items]) if rows_map else None
    xt = _signals(nrows, T, 11, T / 7.0)
    xa = (_signals(items, T, 12, T / 9.0) * 0.8).requires_grad_(True)
    mask = (torch.rand(length, generator=torch.Generator().manual_seed(5)) < 0.5).double() if masked \
        else torch.ones(length, dtype=torch.float64)
    cnt = float(mask.sum())
    w_edr, w_edc = 1.0, 10.0
    sel = rows if rows is not None else torch.arange(items)

    # float64 reference (targets are float32 stores, as on the product path)
    _, Tdb = _edr_db(xt, win)
    Tdb = Tdb.float().double()
    sum_abs = Tdb.abs().sum(dim=[1, 2])
    _, Adb = _edr_db(xa, win)
    edr_item = (Tdb[sel] - Adb).abs().sum(dim=[1, 2]) / sum_abs[sel]
    Tc = orc.db(orc.schroeder(xt[:, start:start + length]), is_squared=True).float().double()
    Ac = orc.db(orc.schroeder(xa[:, start:start + length]), is_squared=True)
    edc_item = ((Tc[sel] - Ac).abs() * mask).sum(-1) / (items * cnt)
    (w_edr * edr_item.sum()).backward()
    g_edr = xa.grad.clone()
    xa.grad = None
    (w_edc * edc_item.sum()).backward()
    g_edc = xa.grad.clone()

    x2 = _pairs(xa.detach())
    Tdb_d, sabs_d, Tc_d = Tdb.float().to(DEV).contiguous(), sum_abs.float().to(DEV), Tc.float().to(DEV).contiguous()
    maskw = (mask / cnt).float().to(DEV) if masked else None
    inv = 1.0 / items if masked else 1.0 / (items * cnt)
    rows_d = None if rows is None else rows.to(DEV)
    gP, part, edc, dxe = ops.decay_items_fwd(x2, items, win, Tdb_d, sabs_d, rows_d, None, w_edr, start, length, Tc_d,
                                             maskw, inv, w_edc, True)
    torch.cuda.synchronize()
    assert rel_err((part[:, 0].cpu().double() / sum_abs[sel]), edr_item.detach()) < 2e-5
    assert rel_err(edc.cpu(), edc_item.detach()) < 2e-5
    # EDC gradient: planar, over the window (sign flips at |diff| ~ 0 are measure-zero: L1 norm)
    ref = g_edc[:, start:start + length]
    assert (dxe.cpu().double() - ref).abs().sum() / ref.abs().sum() < 1e-3
    # EDR gradient: through the STFT adjoint (even launch stores, odd launch adds the planar EDC gradient)
    g = ops.stft_power_pairs_bwd_planar(x2, items, win, gP, 0)
    ops.stft_power_pairs_bwd_planar(x2, items, win, gP, 1, out=g, base=dxe, start=start)
    gx = g.transpose(1, 2).reshape(-1, T)[:items].cpu().double()
    tot = g_edr + g_edc
    assert (gx - tot).abs().sum() / tot.abs().sum() < 1e-3
    # no-gradient mode: same losses, nothing else written
    _, part2, edc2, _ = ops.decay_items_fwd(x2, items, win, Tdb_d, sabs_d, rows_d, None, w_edr, start, length, Tc_d, maskw,
                                            inv, w_edc, False)
    assert torch.equal(part2, part) and torch.equal(edc2, edc)


def test_decay_items_fwd_vs_unfused_kernels(ops):
    """Same inputs through gfdn_stft_power_pairs -> gfdn_edr_loss and gfdn_edc_loss_pairs: losses to float32 rounding,
    gradients in L1; frequency weights on."""
    items, T, start, length, win = 8, 65537, 640, 47360, 4096
    xt = _signals(items, T, 21, T / 7.0).float().to(DEV)
    xa = _signals(items, T, 22, T / 9.0)
    x2 = _pairs(xa)
    wf = (torch.rand(win // 2 + 1, generator=torch.Generator().manual_seed(2)) + 0.5).to(DEV)
    P = ops.stft_power(xt, win)
    Tdb, sabs = ops.edr_target(P)
    Tc = ops.edc_target(xt, start, length)
    maskw = ((torch.rand(length, generator=torch.Generator().manual_seed(3)) < 0.5).float() / (0.5 * length)).to(DEV)
    Pa = ops.stft_power_pairs(x2, items, win)
    li = ops.edr_loss(Pa, Tdb, sabs, wf, gscale=1.5, want_grad=True)          # Pa becomes dL/dP
    lc, g_edc = ops.edc_loss_pairs(x2, items, start, length, Tc, maskw, 1.0 / items, 7.0, True)
    gP, part, edc, dxe = ops.decay_items_fwd(x2, items, win, Tdb, sabs, None, wf, 1.5, start, length, Tc, maskw,
                                             1.0 / items, 7.0, True)
    torch.cuda.synchronize()
    assert rel_err((part[:, 0] / sabs).cpu(), li.cpu()) < 1e-5
    assert rel_err(edc.cpu(), lc.cpu()) < 1e-5
    assert float((gP - Pa).abs().sum() / Pa.abs().sum()) < 2e-4
    ge = g_edc.transpose(1, 2).reshape(-1, T)[:items, start:start + length]
    assert float((dxe - ge).abs().sum() / ge.abs().sum()) < 2e-4
    # the planar odd launch adds exactly what the interleaved one adds
    a = ops.stft_power_pairs_bwd(x2, items, win, Pa, phase=0)
    ops.stft_power_pairs_bwd(x2, items, win, Pa, base=g_edc, out=a, phase=1)
    planar = torch.zeros(items, length, device=DEV)
    planar.copy_(ge)
    b = ops.stft_power_pairs_bwd_planar(x2, items, win, Pa, 0)
    ops.stft_power_pairs_bwd_planar(x2, items, win, Pa, 1, out=b, base=planar, start=start)
    assert torch.equal(a, b)
