"""Same-box probe: the 7-band bank step as ONE bank (one launch per stage for all bands) against the same bands as 2 or 3
independent banks whose graphs replay on separate streams (one bank's latency-bound head and tail beside another's
bandwidth-bound middle).   usage: python tools/split_bank_probe.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from diffgfdn_amd.bandbank import BandBank, BandBankTrainer, BandStackedDataset

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
torch.set_num_threads(4)
centres = bench.BAND_CENTRES
work = []
for q, f in enumerate(centres):
    room, data, net, tc, split, filt, delays = bench.build_workload(dev, 1234 + q, bench.NUM_RECEIVERS, centre_hz=f, room_seed=q,
                                                                   make_trainer=False, t60max=1.5)
    work.append((net, data, filt, split))


def make(bands, seed):
    nets = [work[q][0] for q in bands]
    bank = BandBank(nets)
    tr = BandBankTrainer(bank, bench.trainer_config(500.0, 20, train_dir=f'/tmp/gfdn_bench/split{seed}'),
                         subband_filter_freq_resp=torch.stack([work[q][2] for q in bands]),
                         band_names=[int(centres[q]) for q in bands], data_parallel=False)
    sds = BandStackedDataset([work[q][1] for q in bands])
    sds.precompute_decay_targets(bench.WIN, *tr._target_window(bench.K))
    gen = torch.Generator().manual_seed(seed)
    st = [torch.tensor(work[q][3][0]) for q in bands]
    draw = lambda: sds.global_rows([t[torch.randperm(len(t), generator=gen)[:bench.BATCH]].tolist() for t in st])
    step = tr.graphed(sds, bench.BATCH)
    return step, draw


def run(parts):
    objs = [make(b, 300 + i) for i, b in enumerate(parts)]
    streams = [torch.cuda.Stream() for _ in objs]
    for (step, draw) in objs:
        step.load_schedule([draw() for _ in range(10)])
        for _ in range(10):
            step.run_next()
    torch.cuda.synchronize()
    for (step, draw) in objs:
        step.load_schedule([draw() for _ in range(steps)])
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    t0 = time.perf_counter()
    for _ in range(steps):
        for (step, _), s in zip(objs, streams):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                step.run_next()
    for s in streams:
        main.wait_stream(s)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    print('banks', parts, f'{ms:.4f} ms per 7-band step', flush=True)
    del objs


# NB every BandBank re-views its nets' parameters: one partition at a time, the nets are reused
for parts in ([[0, 1, 2, 3, 4, 5, 6]], [[0, 2, 4, 6], [1, 3, 5]], [[0, 3, 6], [1, 4], [2, 5]], [[0, 1, 2, 3, 4, 5, 6]],
              [[0, 2, 4, 6], [1, 3, 5]]):
    run(parts)
