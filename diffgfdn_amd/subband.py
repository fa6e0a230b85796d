"""Band-parallel driver: one independent GFDN per octave band, bands spread over the GPUs of a node.

Counterpart of the reference's src/run_subband_training_treble.py: ``training`` (:175-204) trains the
bands one after another in one process; ``inferencing`` (:207-375) rebuilds every band's model, renders
the RIRs (``get_response``), filters each with its band's reconstructing FIR (``fftconvolve(h, taps,
'full')``, :321-324) and sums the bands per receiver (:358).

Bands share neither parameters nor data, so on MI355X they are placed on
different ranks with NO collective during training; the only exchange is the final sum over bands
of the filtered RIRs, one ``reduce`` of a (receivers, samples) float32 tensor to rank 0 over
RCCL / xGMI.  FIR taps are an input (the reference takes them from pyfar).
"""
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import hip_ops as ops
from .trainer import get_response


def band_assignment(freqs: Sequence[float], world_size: int) -> List[List[float]]:
    """Round-robin placement of bands on ranks: rank r trains freqs[r::world_size]."""
    return [list(freqs[r::world_size]) for r in range(world_size)]


def train_bands(freqs: Sequence[float], build_band: Callable[[float], tuple], rank: int = 0,
                world_size: int = 1) -> Dict[float, object]:
    """Train this rank's bands one after another (reference ``training``); ``build_band(freq)`` returns
    (trainer, train_loader, valid_loader).  No communication: every band is its own model."""
    trainers = {}
    for f in band_assignment(freqs, world_size)[rank]:
        trainer, train_loader, valid_loader = build_band(f)
        trainer.train(train_loader, valid_loader)
        trainers[f] = trainer
    return trainers


def full_convolve(h: torch.Tensor, taps: torch.Tensor) -> torch.Tensor:
    """scipy.signal.fftconvolve(h[r], taps, mode='full') for every row r, on the library's FFTs:
    zero-pad both to the next power of two >= len(h) + len(taps) - 1, multiply spectra, invert."""
    T, M = h.shape[-1], taps.numel()
    full = T + M - 1
    n = 1 << (full - 1).bit_length()
    Hf = ops.rfft_pow2(h, n)
    Tf = ops.rfft_pow2(taps.reshape(1, -1).to(h.device), n)
    y = ops.irfft_pow2_fwd(Hf * Tf, n)
    return y[..., :full]


@torch.no_grad()
def render_band(net, batches, taps: torch.Tensor) -> torch.Tensor:
    """Filtered RIRs of one band for all batches (reference :308-324): (receivers, nfft + taps - 1)."""
    out = []
    for data in batches:
        h = get_response(data, net)[-1]
        out.append(full_convolve(h.contiguous(), taps))
    return torch.cat(out, dim=0)


@torch.no_grad()
def infer_bands(freqs: Sequence[float], load_band: Callable[[float], tuple], rank: int = 0, world_size: int = 1,
                group=None, dst: int = 0) -> Optional[torch.Tensor]:
    """The reference's ``inferencing`` (:207-375) over the ranks of a job: this rank renders ITS bands
    (``load_band(freq)`` -> (net with the trained state loaded, batches, FIR taps of the band's reconstructing
    filter)), filters them (``render_band``) and the bands are summed per receiver on ``dst`` (``sum_bands``; a rank
    without bands contributes zeros).  Returns the full-band RIRs (receivers, nfft + taps - 1) on ``dst``."""
    local, device = [], None
    for f in band_assignment(freqs, world_size)[rank]:
        net, batches, taps = load_band(f)
        net.eval()
        local.append(render_band(net, batches, taps))
        device = local[-1].device
    return sum_bands(local, group=group, dst=dst, device=device)


def sum_bands(local_band_rirs: Sequence[torch.Tensor], group=None, dst: int = 0, device=None) -> Optional[torch.Tensor]:
    """Sum of the filtered RIRs over ALL bands (reference :358 ``groupby('position').apply(sum)``):
    local sum over this rank's bands, then one reduce to ``dst``.  Returns the total on ``dst``,
    None elsewhere.  A rank WITHOUT bands (8 ranks, 7 bands) passes []: the shape, dtype of the result are agreed
    on first (an all-reduce MAX of a 3-word header) and the rank contributes zeros; ``device`` is where it builds
    them (default: the current CUDA device, or the CPU when there is none)."""
    local = list(local_band_rirs)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        if not local:
            raise ValueError("sum_bands: no bands")
        return torch.stack(local, dim=0).sum(dim=0)
    if local:
        total = torch.stack(local, dim=0).sum(dim=0)
        if total.dim() != 2:
            raise ValueError("sum_bands: (receivers, samples) tensors expected")
        device = total.device
    else:
        total = None
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    dtypes = [torch.float32, torch.float64]
    head = torch.tensor([0, 0, 0] if total is None else [total.shape[0], total.shape[1], dtypes.index(total.dtype)],
                        dtype=torch.int64, device=device)
    dist.all_reduce(head, op=dist.ReduceOp.MAX, group=group)
    shape, dtype = (int(head[0]), int(head[1])), dtypes[int(head[2])]
    if total is None:
        if shape[0] == 0:
            raise ValueError("sum_bands: no rank holds a band")
        total = torch.zeros(shape, dtype=dtype, device=device)
    elif tuple(total.shape) != shape:
        raise ValueError("sum_bands: the ranks' band responses differ in shape")
    dist.reduce(total, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return total if dist.get_rank(group) == dst else None
