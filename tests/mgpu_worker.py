"""Worker of tests/test_gpu_multi.py: one rank of an N-rank RCCL job on N GPUs (started by torch.distributed.run).
Every rank steps its shard of the band bank's batch; checks (rank 0 prints OK when all hold on all ranks):
  (1) the all-reduced gradient bucket equals the sum of the ranks' own buckets (gathered for the check);
  (2) the graph-replayed data-parallel step -- the collective captured inside the graph when every rank can, else two graphs
      with the eager all-reduce between them -- equals the host-launched step bit for bit (parameters, Adam moments, losses);
  (3) the ranks hold identical parameters afterwards."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    # MGPU_BACKEND=gloo: a rehearsal of this script's logic with the ranks sharing the visible devices (host-side collective)
    backend = os.environ.get("MGPU_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from tests import test_gpu_bank as tb
    sel = [[0, 3, 5, 7], [1, 2, 8, 11], [4, 6, 9, 10]]
    B = 4 // world if 4 % world == 0 else 4
    mine = [s[rank * B:(rank + 1) * B] for s in sel] if 4 % world == 0 else sel
    ok = True
    res = {}
    for mode in ("eager", "graph"):
        nets, data, filt, bank, tr, sds, (start, length) = tb._bank_setup(mask=True)
        assert tr.world_size == world and tr._allreduce is not None
        tr.allreduce_in_graph = True
        step = tr.graphed(sds, len(mine[0]), mask_seed=7)
        rows = sds.global_rows(mine)
        if mode == "graph":
            step.capture(rows)
            out = step(rows)
        else:
            # the same launch sequence from the host: explicit step up to the gradients, the collective, the update
            step._load_inputs(rows)
            own = None
            orig = tr._allreduce

            def checked():
                nonlocal own
                own = tr.optimizer.bucket.detach().clone()
                orig()
            tr._allreduce = checked
            out = step._eager()
            tr._allreduce = orig
            torch.cuda.synchronize()
            gathered = [torch.empty_like(own) for _ in range(world)]
            dist.all_gather(gathered, own)
            want = gathered[0].double()
            for g in gathered[1:]:
                want = want + g.double()
            got = tr.optimizer.bucket.detach().double()
            dev = float((got - want).abs().max() / (want.abs().max() + 1e-30))
            ok &= dev < 1e-6
            if rank == 0:
                print(f"bucket all-reduce vs sum of {world} rank buckets: max deviation {dev:.2e}", flush=True)
        torch.cuda.synchronize()
        res[mode] = ({k: v.detach().cpu().numpy().copy() for k, v in out.items()},
                     tr.optimizer.flat_param.detach().clone(), tr.optimizer.exp_avg.detach().clone(),
                     getattr(step, "allreduce_in_graph", None), getattr(step, "collective_probe_nodes", None))
    for k, v in res["eager"][0].items():
        ok &= bool(np.allclose(res["graph"][0][k], v, rtol=1e-6, atol=0))
    ok &= bool(torch.equal(res["graph"][1], res["eager"][1])) and bool(torch.equal(res["graph"][2], res["eager"][2]))
    # identical parameters on every rank
    p = res["graph"][1]
    ref = p.clone()
    dist.broadcast(ref, src=0)
    ok &= bool(torch.equal(p, ref))
    flag = torch.tensor([1 if ok else 0], device="cuda", dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"captured all-reduce: {res['graph'][3]} ({res['graph'][4]} node(s) in the capture probe)", flush=True)
        print("MGPU_OK" if int(flag.item()) == 1 else "MGPU_FAIL", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
