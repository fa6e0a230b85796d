"""CPU-oracle steps of the full-size tests as jobs for worker processes -- TEST INFRASTRUCTURE.

The seven band steps of tests/test_gpu_fullsize.py::test_full_size_bench_shape_vs_oracle are independent float64 evaluations
of the reference's step (oracle/cpu_trainer.OracleGridTrainer), each 8-15 s on 16 threads; run one after the other they were
a quarter of the GPU suite's time.  Here they run side by side in a few spawned worker processes (CPU only: a worker
never touches the GPU) with the threads divided between them.  Inputs and outputs are plain CPU tensors / numpy arrays."""
import os
from concurrent.futures import ProcessPoolExecutor
from multiprocessing import get_context


def grid_step(job):
    """One normalize + train_step of the CPU oracle.  job: dict(fs, delays, G, sd (state dict of CPU tensors),
    common_decay_times, n_fourier, filt (complex128 or None), batch (dict of CPU tensors), keep (index tensor or None),
    threads).  Returns (parts, grads, after) keyed like the model's state dict (numpy arrays)."""
    import torch
    torch.set_num_threads(int(job.get("threads", 4)))
    from oracle import gfdn_oracle as orc
    from oracle.cpu_trainer import OracleGridTrainer
    sd = job["sd"]
    lin, norm, names = [], [], []
    for i in range(64):
        k = f"output_scalars.mlp.model.{i}.weight"
        if k in sd:
            pair = (sd[k].clone(), sd[f"output_scalars.mlp.model.{i}.bias"].clone())
            (lin if sd[k].ndim == 2 else norm).append(pair)
            names.append((f"output_scalars.mlp.model.{i}", pair))
    p = orc.GridModelParams(job["fs"], job["delays"], job["G"], sd["input_gains"].clone(), sd["output_gains"].clone(),
                            sd["feedback_loop.M"].clone(), sd["feedback_loop.alpha"].clone(), job["common_decay_times"],
                            lin, norm, job["n_fourier"])
    otr = OracleGridTrainer(p, lr=1e-3, io_lr=1e-2, edr_weight=1.0, edc_weight=10.0, spectral_weight=1.0,
                            sparsity_weight=2.0, use_asym=True, subband_filter=job["filt"])
    otr.normalize(job["batch"])
    _, parts = otr.train_step(job["batch"], job["keep"])
    grads = {"input_gains": p.input_gains.grad, "output_gains": p.output_gains.grad, "feedback_loop.M": p.M.grad}
    after = {"input_gains": p.input_gains.detach(), "output_gains": p.output_gains.detach(), "feedback_loop.M": p.M.detach()}
    for base, (w, bias) in names:
        grads[base + ".weight"], grads[base + ".bias"] = w.grad, bias.grad
        after[base + ".weight"], after[base + ".bias"] = w.detach(), bias.detach()
    return ({k: float(v) for k, v in parts.items()}, {k: v.clone() for k, v in grads.items()},
            {k: v.clone() for k, v in after.items()})


def run_grid_steps(jobs, workers: int = 4):
    """The jobs' results in order.  ``workers`` processes (spawned: the parent holds a GPU context that must not be
    forked), the host's cores divided between them."""
    if len(jobs) <= 1 or workers <= 1:
        return [grid_step(j) for j in jobs]
    workers = min(workers, len(jobs))
    threads = max(1, min(16, os.cpu_count() or 1) // workers)
    for j in jobs:
        j["threads"] = threads
    with ProcessPoolExecutor(max_workers=workers, mp_context=get_context("spawn")) as pool:
        return list(pool.map(grid_step, jobs))
