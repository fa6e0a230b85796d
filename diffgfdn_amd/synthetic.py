"""Synthetic multi-slope room data of the shape the reference trains on.

No dataset ships with the reference tree (Git-LFS pointers only), so the benchmark and the parity
tests draw RIRs  h_r[t] = sum_g a_{r,g} exp(-6.908 t / (fs T60_g)) w_r[t],  w ~ N(0,1),  for R
receivers on a 10 m x 13 m floor, one source.  Deterministic in ``seed``."""
from typing import Dict

import numpy as np


def synthetic_room(num_receivers: int = 838, num_groups: int = 4, sample_rate: float = 32000.0,
                   rir_len: int = 64000, seed: int = 0, t60_range=(0.3, 1.5)) -> Dict:
    rng = np.random.RandomState(seed)
    T60 = np.linspace(t60_range[0], t60_range[1], num_groups)
    t = np.arange(rir_len, dtype=np.float64)
    env = np.exp(-6.908 * t[None, :] / (sample_rate * T60[:, None]))          # (G, T)
    amps = rng.uniform(0.1, 1.0, (num_receivers, num_groups))
    rirs = np.empty((num_receivers, rir_len), dtype=np.float64)
    for r in range(num_receivers):                                             # bounded memory
        rirs[r] = (amps[r] @ env) * rng.randn(rir_len)
    pos = np.stack([rng.uniform(0, 10, num_receivers), rng.uniform(0, 13, num_receivers),
                    np.full(num_receivers, 1.5)], axis=1)
    return {'sample_rate': sample_rate, 'rirs': rirs, 'receiver_position': pos,
            'source_position': np.array([2.0, 3.0, 1.5]), 'common_decay_times': T60[None, :],
            'amplitudes': amps, 'num_rooms': num_groups}
