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
    """The trainer's weighting (diffgfdn_amd/trainer.py::_step_losses) with oracle losses."""
    fs = p.sample_rate
    H, Hs = orc.grid_model_forward(p, batch)
    tgt = batch["target_rir_response"]
    B = H.shape[0]
    edr = orc.edr_loss(tgt, H, int(fx["win"]), int(fx["hop"]))                 # sum over local items
    L = orc.ms_to_samps(float(np.max(p.common_decay_times)) * 1e3, fs)
    edc_local_mean = orc.edc_loss(tgt, H, L, orc.ms_to_samps(20.0, fs))        # mean over local items
    edc = edc_local_mean * B / global_batch                                    # -> share of the global mean
    crit = orc.amse_loss if asym else orc.mse_loss
    spec = sum(crit(Hs[0][..., k], torch.ones_like(Hs[0][..., k])) for k in range(p.num_groups))
    spars = orc.sparsity_loss(orc.ortho_param(p.M[p.num_groups - 1]))
    return 1.0 * edr + 10.0 * edc + (1.0 * spec + 2.0 * spars) / world


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
