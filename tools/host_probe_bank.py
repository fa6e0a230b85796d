"""Band bank: host-side cost of one graphed step next to the GPU's (is replay throughput host- or GPU-bound?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda', 0)
centres = bench.BAND_CENTRES
(room, delays, filt), data, net, trainer, splits = bench.build_bank_workload(dev, 1234, centres, bench.NUM_RECEIVERS)
step = trainer.graphed(data, bench.BATCH)
gen = torch.Generator().manual_seed(100)
splits_t = [torch.tensor(s[0]) for s in splits]
def draw():
    sel = [t[torch.randperm(len(t), generator=gen)[:bench.BATCH]].tolist() for t in splits_t]
    return data.global_rows(sel)
for _ in range(20): step(draw())
torch.cuda.synchronize()
N = 400
acc = {'draw': 0.0, 'load': 0.0, 'replay': 0.0}
t_all = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter(); sel = draw()
    t1 = time.perf_counter(); step._load_inputs(sel)
    t2 = time.perf_counter(); step.graph_a.replay()
    t3 = time.perf_counter()
    acc['draw'] += t1 - t0; acc['load'] += t2 - t1; acc['replay'] += t3 - t2
t_host = time.perf_counter() - t_all
torch.cuda.synchronize()
t_tot = time.perf_counter() - t_all
print(f"wall/step {t_tot/N*1e3:.4f} ms; host loop/step {t_host/N*1e3:.4f} ms;", {k: round(v / N * 1e3, 4) for k, v in acc.items()})
sel = draw(); step._load_inputs(sel)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): step.graph_a.replay()
t_h = time.perf_counter() - t0
torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print(f"replay only (no index copy between): wall/step {t1/N*1e3:.4f} ms, host/replay {t_h/N*1e3:.4f} ms")
sels = [draw() for _ in range(N)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in sels:
    step._load_inputs(s); step.graph_a.replay()
t_h = time.perf_counter() - t0
torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print(f"load + replay (receivers drawn beforehand): wall/step {t1/N*1e3:.4f} ms, host {t_h/N*1e3:.4f} ms")
