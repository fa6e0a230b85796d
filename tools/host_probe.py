"""Host-side cost of one graphed step: is replay throughput host- or GPU-bound? (diagnostic)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(4)
import bench
dev = torch.device('cuda', 0)
room, data, net, trainer, train_idx, filt, delays = bench.build_workload(dev, 1234, 838)
step = trainer.graphed(data, bench.BATCH)
gen = torch.Generator().manual_seed(100)
tt = torch.tensor(train_idx)
def draw():
    return tt[torch.randperm(len(train_idx), generator=gen)[:bench.BATCH]].tolist()
for _ in range(20): step(draw())
torch.cuda.synchronize()
N = 1000
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
print(f"wall/step {t_tot/N*1e3:.3f} ms; host loop/step {t_host/N*1e3:.3f} ms;", {k: round(v / N * 1e3, 4) for k, v in acc.items()})
# GPU-only: replay the same inputs back-to-back
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): step.graph_a.replay()
t_h = time.perf_counter() - t0
torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print(f"replay only: wall/step {t1/N*1e3:.3f} ms, host/replay {t_h/N*1e3:.3f} ms")
# single replay latency (sync each)
ts = []
for _ in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step.graph_a.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort(); print(f"one replay, synced: median {ts[25]*1e3:.3f} ms min {ts[0]*1e3:.3f}")
# finer split of _load_inputs while a replay is in flight
import collections
acc = collections.defaultdict(float)
tr = trainer; crit = tr.criterion[1]
for _ in range(N):
    sel = draw()
    host_mask, host_idx, ev = step._ring[step._ring_pos]
    step._ring_pos = (step._ring_pos + 1) % len(step._ring)
    t0 = time.perf_counter(); ev.synchronize()
    t1 = time.perf_counter(); host_idx.copy_(torch.as_tensor(list(sel), dtype=torch.long))
    t2 = time.perf_counter(); step.idx.copy_(host_idx, non_blocking=True)
    t3 = time.perf_counter(); keep = torch.bernoulli(torch.empty(step.length).uniform_(0, 1))
    t4 = time.perf_counter(); torch.div(keep, float(keep.sum()) * step.gb, out=host_mask)
    t5 = time.perf_counter(); step.maskw.copy_(host_mask, non_blocking=True)
    t6 = time.perf_counter(); ev.record()
    t7 = time.perf_counter(); step.graph_a.replay()
    t8 = time.perf_counter()
    for k, v in zip(['evsync', 'idxhost', 'idxh2d', 'bern', 'div', 'maskh2d', 'evrec', 'replay'],
                    [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7]):
        acc[k] += v
torch.cuda.synchronize()
print({k: round(v / N * 1e3, 4) for k, v in acc.items()})
