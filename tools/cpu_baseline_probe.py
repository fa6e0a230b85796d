import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nt = int(sys.argv[1])
torch.set_num_threads(nt)
import bench
from diffgfdn_amd.synthetic import synthetic_room
from diffgfdn_amd.config import DiffGFDNConfig
room = synthetic_room(64, 4, bench.FS, 64000, seed=0)
delays = DiffGFDNConfig(num_groups=4, num_delay_lines=16, sample_rate=bench.FS, seed=23963).delay_length_samps
filt = bench.octave_band_response(500.0, bench.FS, bench.NFFT)
t0 = time.time()
orig = torch.set_num_threads
torch.set_num_threads = lambda n: None     # keep the probe's thread count
r = bench.cpu_baseline(room, delays, filt, steps=1)
print(nt, 'threads:', r['sec_per_step'], 's/step; total', time.time() - t0, flush=True)
