import os, sys, cProfile, pstats, io, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda', 0)
print('torch threads', torch.get_num_threads(), flush=True)
room, data, net, trainer, train_idx, filt, delays = bench.build_workload(dev, 1234, int(os.environ.get('R', 128)))
sel = train_idx[:32]
def whole():
    b = data.collate(sel, lean=True); trainer.normalize(b); trainer.train_step(b)
for _ in range(3): whole()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): whole()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25); print(s.getvalue()[:6000])
