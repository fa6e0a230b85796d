"""World-size-2 gloo tests (CPU) of the data-parallel path: receiver shards per rank, one flat
all-reduce of the gradients, and the loss weighting that makes the sharded sum equal the
single-process gradient.  The HIP kernels need a GPU, so the per-rank compute here is the
CPU oracle; the sharding (GridLoader), the flat all-reduce (FlatGradAllReduce) and the weighting rules
are the product's."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gfdn_oracle as orc
from tests.helpers import batch_from, load
from tests.test_oracle_golden import grid_params


def _local_loss(p, batch, fx, world, global_batch, asym):
    """A rank's share of the step loss: the oracle's LOCAL sums times the factors the trainers themselves use
    (diffgfdn_amd.losses.shard_loss_scales -- VarReceiverPosTrainer._step_losses, FusedBankStep.run and decay_losses
    take theirs from the same function)."""
    from diffgfdn_amd.losses import shard_loss_scales
    fs = p.sample_rate
    H, Hs = orc.grid_model_forward(p, batch)
    tgt = batch["target_rir_response"]
    B = H.shape[0]
    edr = orc.edr_loss(tgt, H, int(fx["win"]), int(fx["hop"]))                 # sum over local items
    L = orc.ms_to_samps(float(np.max(p.common_decay_times)) * 1e3, fs)
    mix = orc.ms_to_samps(20.0, fs)
    count = min(L, tgt.shape[-1]) - mix                                        # kept time indices (no mask: all)
    scale = shard_loss_scales(world, global_batch, count)
    edc_local_sum = orc.edc_loss(tgt, H, L, mix) * B * count                   # sum of |dB| over local items x indices
    crit = orc.amse_loss if asym else orc.mse_loss
    spec = sum(crit(Hs[0][..., k], torch.ones_like(Hs[0][..., k])) for k in range(p.num_groups))
    spars = orc.sparsity_loss(orc.ortho_param(p.M[p.num_groups - 1]))
    return (1.0 * scale["edr"] * edr + 10.0 * scale["edc"] * edc_local_sum
            + scale["colorless"] * (1.0 * spec + 2.0 * spars))


def _params_list(p):
    out = [p.input_gains, p.output_gains, p.M]
    for w, b in list(p.mlp_weights) + list(p.mlp_norms):
        out += [w, b]
    return out


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from diffgfdn_amd.dataloader import GridLoader
    from diffgfdn_amd.trainer import FlatGradAllReduce
    fx = load("f234_n12_k257.npz")
    full = batch_from(fx)
    B = full["target_rir_response"].shape[0]

    class _DS:                       # index-list dataset: collate returns the shard's rows
        def collate(self, idx, lean=False):
            idx = list(idx)
            return {k: (v if k == "z_values" else v[idx]) for k, v in full.items()}

    loader = GridLoader(_DS(), list(range(B)), batch_size=B, shuffle=False, rank=rank, world_size=world)
    shard = next(iter(loader))
    assert shard["target_rir_response"].shape[0] == B // world
    p = grid_params(fx, requires_grad=True)
    loss = _local_loss(p, shard, fx, world, B, True)
    loss.backward()
    params = _params_list(p)
    FlatGradAllReduce(params)()
    if rank == 0:
        ret["grads"] = [q.grad.clone() for q in params]
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gradients_equal_single_process():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    fx = load("f234_n12_k257.npz")
    p = grid_params(fx, requires_grad=True)
    full = batch_from(fx)
    loss = _local_loss(p, full, fx, 1, full["target_rir_response"].shape[0], True)
    loss.backward()
    for got, q in zip(ret["grads"], _params_list(p)):
        err = float((got - q.grad).abs().max() / (q.grad.abs().max() + 1e-30))
        assert err < 1e-5, err


def _allreduce_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffgfdn_amd.trainer import FlatGradAllReduce
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    x = torch.arange(40, dtype=torch.float32).reshape(8, 5) / 10.0
    net(x[rank::world]).pow(2).sum().backward()
    net[2].bias.grad = None                      # a parameter without gradient on this rank
    FlatGradAllReduce(net.parameters())()
    if rank == 0:
        ret["g"] = [q.grad.clone() for q in net.parameters()]
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_sums_over_ranks():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_allreduce_worker, args=(world, port, ret), nprocs=world, join=True)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    x = torch.arange(40, dtype=torch.float32).reshape(8, 5) / 10.0
    net(x).pow(2).sum().backward()
    ps = list(net.parameters())
    for i, (got, q) in enumerate(zip(ret["g"], ps)):
        if i == 3:
            assert float(got.abs().max()) == 0.0          # missing grads enter the sum as zeros
        else:
            assert torch.allclose(got, q.grad, rtol=1e-5, atol=1e-6)


def _band_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffgfdn_amd.subband import band_assignment, sum_bands
    freqs = [63, 125, 250, 500, 1000, 2000, 4000]
    mine = band_assignment(freqs, world)[rank]
    # "filtered RIRs" of a band: a deterministic function of the band, (receivers=3, samples=8)
    local = [torch.full((3, 8), float(f)) + torch.arange(8.0) for f in mine]
    total = sum_bands(local)
    if rank == 0:
        ret["total"] = total
        ret["mine"] = mine
    dist.barrier()
    dist.destroy_process_group()


def test_band_parallel_assignment_and_sum():
    from diffgfdn_amd.subband import band_assignment
    freqs = [63, 125, 250, 500, 1000, 2000, 4000]
    parts = band_assignment(freqs, 8)
    assert sorted(f for p in parts for f in p) == freqs and len(parts) == 8 and parts[7] == []
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_band_worker, args=(world, port, ret), nprocs=world, join=True)
    expect = sum(torch.full((3, 8), float(f)) + torch.arange(8.0) for f in freqs)
    assert torch.equal(ret["total"], expect)
    assert ret["mine"] == freqs[0::2]


# ---------------------------------------------------------------------------------------------------------------
# equal step counts on every rank (ragged tails), epoch aggregates, bandless ranks, the bank's bucket / mask code
# ---------------------------------------------------------------------------------------------------------------
def test_rank_shards_are_equal_and_cover_full_batches():
    from diffgfdn_amd.bandbank import bank_shards
    from diffgfdn_amd.dataloader import GridLoader, rank_shard
    for world in (2, 3, 8):
        for n in range(0, 40):
            chunk = list(range(100, 100 + n))
            shares = [rank_shard(chunk, r, world) for r in range(world)]
            assert len({len(s) for s in shares}) == 1                       # the same size on every rank
            flat = sorted(i for s in shares for i in s)
            assert flat == chunk[:(n // world) * world]                     # disjoint, only the ragged rest dropped
    # loader: tails smaller than the world are skipped on ALL ranks; every rank yields the same number of batches

    class _DS:
        def collate(self, idx, lean=False):
            return list(idx)

    for world in (2, 8):
        for n in (64, 65, 66, 67, 71, 72, 3):
            per_rank = [list(GridLoader(_DS(), list(range(n)), batch_size=32, shuffle=False, rank=r, world_size=world))
                        for r in range(world)]
            assert len({len(b) for b in per_rank}) == 1, (world, n)
            assert all(len(b) == len(GridLoader(_DS(), list(range(n)), 32, False, rank=0, world_size=world))
                       for b in per_rank)
            for step in zip(*per_rank):
                assert len({len(s) for s in step}) == 1 and len(step[0]) > 0
    # bank: per band the same receivers as the single-process batch, split over the ranks
    orders = [list(range(10, 48)), list(range(60, 98))]                     # 38 receivers per band, batch 16
    for world in (1, 2, 4):
        steps = [list(bank_shards(orders, 16, r, world)) for r in range(world)]
        assert len({len(s) for s in steps}) == 1
        for i, per_rank in enumerate(zip(*steps)):
            for q in range(2):
                got = sorted(x for sel in per_rank for x in sel[q])
                full = orders[q][16 * i:16 * (i + 1)]
                assert got == full[:(len(full) // world) * world]


def _epoch_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from diffgfdn_amd.bandbank import BandBankTrainer
    from diffgfdn_amd.bankstep import FusedBankStep
    from diffgfdn_amd.losses import edc_loss
    from diffgfdn_amd.subband import band_assignment, sum_bands
    from diffgfdn_amd.trainer import reduce_epoch_losses
    # (1) epoch aggregates: decay terms add up over the ranks, colorless terms are the same everywhere
    agg = {"edr_loss": torch.tensor([1.0 + rank, 2.0]), "edc_loss": torch.tensor([0.5, 0.25 * (rank + 1)]),
           "spectral_loss": torch.tensor([3.0, 4.0]), "sparsity_loss": torch.tensor([0.1, 0.2])}
    red = reduce_epoch_losses(agg)
    # (2) 8 bands-less-one style: rank 1 holds no band here
    freqs = [63, 125, 250]
    mine = band_assignment(freqs, world)[rank] if rank == 0 else []
    mine = freqs if rank == 0 else []
    total = sum_bands([torch.full((3, 8), float(f)) for f in mine], device=torch.device("cpu"))
    # (3) the bank trainer's mask / weighting code on a stand-in trainer: same mask on every rank, 1 / (B_global count)
    fake = SimpleNamespace(criterion=[None, edc_loss(1500.0, 8000.0, use_mask=True)], world_size=world,
                           process_group=None)
    torch.manual_seed(100 + rank)                                             # ranks draw DIFFERENT masks ...
    maskw, inv = BandBankTrainer._draw_edc_mask(fake, 640, 4, torch.device("cpu"))
    # (4) the explicit step's bucket: loss slots behind the gradients, one all-reduce, totals rebuilt from the slots
    nb, n = 2, 5
    bucket = torch.zeros(n + 3 * nb)
    bucket[:n] = torch.arange(n, dtype=torch.float32) * (rank + 1)
    bucket[n:].view(3, nb).copy_(torch.tensor([[1.0, 2.0], [10.0, 20.0], [0.5, 0.25]]) * (rank + 1))
    stepped = []
    opt = SimpleNamespace(bucket=bucket, extra=bucket[n:], flat_grad=bucket[:n], step=lambda: stepped.append(1))
    tr = SimpleNamespace(num_bands=nb, optimizer=opt)
    sums, tot = FusedBankStep.finish(SimpleNamespace(tr=tr), lambda: dist.all_reduce(opt.bucket))
    ret[rank] = {"red": {k: v.clone() for k, v in red.items()}, "total": total, "mask": maskw.clone(), "inv": inv,
                 "grad": bucket[:n].clone(), "sums": sums.clone(), "tot": tot.clone(), "stepped": len(stepped)}
    dist.barrier()
    dist.destroy_process_group()


def test_epoch_reduction_bandless_rank_mask_and_bucket():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_epoch_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    for k in r0["red"]:
        assert torch.equal(r0["red"][k], r1["red"][k]), k                  # identical on every rank
    assert torch.equal(r0["red"]["edr_loss"], torch.tensor([3.0, 4.0]))      # (1 + 2, 2 + 2)
    assert torch.equal(r0["red"]["edc_loss"], torch.tensor([1.0, 0.75]))
    assert torch.equal(r0["red"]["spectral_loss"], torch.tensor([3.0, 4.0]))
    assert torch.allclose(r0["red"]["sparsity_loss"], torch.tensor([0.1, 0.2]))
    assert torch.equal(r0["total"], torch.full((3, 8), 63.0 + 125.0 + 250.0)) and r1["total"] is None
    assert torch.equal(r0["mask"], r1["mask"])                               # rank 0's draw wins
    count = float(r0["mask"].sum())
    assert r0["inv"] == r1["inv"] == 1.0 / (4 * world * count)
    assert torch.equal(r0["grad"], torch.arange(5, dtype=torch.float32) * 3) and torch.equal(r0["grad"], r1["grad"])
    assert torch.equal(r0["sums"][:, 1], torch.tensor([3.0, 6.0])) and torch.equal(r0["sums"][:, 2], torch.tensor([30.0, 60.0]))
    assert torch.equal(r0["tot"], torch.tensor([3.0 + 30.0 + 1.5, 6.0 + 60.0 + 0.75]))
    assert r0["stepped"] == r1["stepped"] == 1


def _agree_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from diffgfdn_amd.trainer import GraphedTrainStep
    me = SimpleNamespace(tr=SimpleNamespace(process_group=None, rank=rank), idx=torch.zeros(1, dtype=torch.long))
    # rank 1 "cannot capture": BOTH ranks must end on the two-graph structure; all able: both capture
    ret[f"mixed{rank}"] = GraphedTrainStep._all_ranks_agree(me, rank == 0)
    ret[f"all{rank}"] = GraphedTrainStep._all_ranks_agree(me, True)
    ret[f"none{rank}"] = GraphedTrainStep._all_ranks_agree(me, False)
    dist.barrier()
    dist.destroy_process_group()


def test_step_structure_is_agreed_over_the_group():
    """One rank failing the capture probe takes every rank to the two-graph step (a captured all-reduce on one rank
    against an eager one on another would hang): GraphedTrainStep._all_ranks_agree is a MIN over the group."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + ((os.getpid() + 777) % 2000)
    mp.spawn(_agree_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[f"mixed{r}"] is False and ret[f"all{r}"] is True and ret[f"none{r}"] is False
