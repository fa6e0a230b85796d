"""Dataset front end: RIRs -> frequency responses on the z grid, resident in HBM.

Mirror of the reference's src/diff_gfdn/dataloader.py for the grid-of-receivers case:
RoomDataset.__init__ (:188-254: rfft of the RIRs), early_late_split (:300-325: 20 ms mixing time,
5 ms Hann fades applied IN PLACE on views of ``rirs`` after the full-RIR rfft), MultiRIRDataset
(:515-600: z = polar(r, 2 pi rfftfreq(nfft))), custom_collate (:674-704: batch dict), load_dataset
(:780-867: fixed 10 % test split with its own generator, then random train/valid split).

MI355X layout: one process keeps the whole grid on its GPU -- 838 receivers x 65 537 bins of
complex64 for the direct path (0.44 GB) plus, once computed, the model-independent target EDR
(838 x 32 x 2049 f32, 0.22 GB) and EDC (838 x 47 360 f32, 0.16 GB); a batch is an index list.
The three load-time rFFTs run on the library's own power-of-two real FFT (csrc/pow2.hip,
``gfdn_rfft_pow2``) in float32 -> complex64, which is the precision every consumer works in.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip_ops as ops


def ms_to_samps(ms: float, fs: float) -> int:
    return int(ms * 1e-3 * fs)


class RoomDataset:
    """RIR grid of one (coupled) room, reference dataloader.py:185-422."""

    def __init__(self, num_rooms: int, sample_rate: float, source_position: np.ndarray,
                 receiver_position: np.ndarray, rirs: np.ndarray, common_decay_times,
                 room_dims=None, room_start_coord=None, band_centre_hz=None, amplitudes=None,
                 noise_floor=None, absorption_coeffs=None, aperture_coords=None,
                 mixing_time_ms: float = 20.0, nfft: Optional[int] = None,
                 grid_spacing_m: float = 0.3, device: str = 'cuda', chunk: int = 128):
        self.sample_rate = sample_rate
        self.num_rooms = num_rooms
        self.source_position = np.asarray(source_position)
        self.receiver_position = np.asarray(receiver_position)
        self.rirs = rirs
        self.band_centre_hz = band_centre_hz
        self.common_decay_times = np.asarray(common_decay_times)
        self.amplitudes = amplitudes
        self.noise_floor = noise_floor
        self.num_rec = self.receiver_position.shape[0]
        self.num_src = self.source_position.shape[0] if self.source_position.ndim > 1 else 1
        self.rir_length = self.rirs.shape[-1]
        self.mixing_time_ms = mixing_time_ms
        self.nfft = nfft
        self._eps = 1e-12
        self.device = device
        self._front_end(chunk)

    @property
    def num_freq_bins(self) -> int:
        if self.nfft is not None:
            return self.nfft
        max_rt60_samps = self.common_decay_times.max() * self.sample_rate
        return int(np.power(2, np.ceil(np.log2(max_rt60_samps))))

    @property
    def norm_receiver_position(self) -> np.ndarray:
        p = self.receiver_position
        lo, hi = p.min(axis=0), p.max(axis=0)
        return (p - lo) / ((hi - lo) + self._eps)

    @property
    def freq_bins_rad(self) -> np.ndarray:
        return np.fft.rfftfreq(self.num_freq_bins) * 2 * np.pi

    def _front_end(self, chunk: int):
        """rfft(full) -> in-place fades -> rfft(late), rfft(early)  (reference :250, :300-325)."""
        nfft = self.num_freq_bins
        K = nfft // 2 + 1
        fs = self.sample_rate
        mix = ms_to_samps(self.mixing_time_ms, fs)
        win = ms_to_samps(5.0, fs)
        w = np.hanning(win)
        R = self.num_rec
        dev = self.device
        self.rir_mag_response = torch.empty((R, K), dtype=torch.complex64, device=dev)
        self.early_rir_mag_response = torch.empty((R, K), dtype=torch.complex64, device=dev)
        self.late_rir_mag_response = torch.empty((R, K), dtype=torch.complex64, device=dev)
        # the early responses in the TIME domain, float64 (5 MB at 838 receivers): what the band bank's float64 build of its
        # direct-path store starts from (BandStackedDataset.direct_time; the reference's rfft runs in float64, :250)
        self.early_rir_time = torch.empty((R, mix), dtype=torch.float64, device=dev)
        fade_out = torch.tensor(w[win // 2:], device=dev)
        fade_in = torch.tensor(w[:win // 2], device=dev)
        rfft = lambda t: ops.rfft_pow2(t.to(torch.float32).contiguous(), nfft)
        for r0 in range(0, R, chunk):
            blk = torch.as_tensor(self.rirs[r0:r0 + chunk], dtype=torch.float64, device=dev)
            self.rir_mag_response[r0:r0 + chunk] = rfft(blk)
            # the reference windows views of self.rirs in place AFTER the full-RIR rfft
            ko = win - win // 2                   # reference slices [-win // 2:] == ceil(win / 2)
            blk[:, mix - ko:mix] *= fade_out
            blk[:, mix:mix + win // 2] *= fade_in
            self.late_rir_mag_response[r0:r0 + chunk] = rfft(blk[:, mix:])
            self.early_rir_mag_response[r0:r0 + chunk] = rfft(blk[:, :mix])
            self.early_rir_time[r0:r0 + chunk] = blk[:, :mix]
            if isinstance(self.rirs, np.ndarray):
                self.rirs[r0:r0 + chunk] = blk.cpu().numpy()     # keep the in-place side effect


class MultiRIRDataset(torch.utils.data.Dataset):
    """Frequency-domain grid dataset, reference dataloader.py:515-600; tensors live on the GPU."""

    def __init__(self, device, room_data: RoomDataset, new_sampling_radius: Optional[float] = None):
        self.device = device
        self.source_position = torch.tensor(np.atleast_2d(room_data.source_position), device=device)
        self.listener_positions = torch.tensor(room_data.receiver_position, device=device)
        self.norm_listener_position = torch.tensor(room_data.norm_receiver_position, device=device)
        w = torch.tensor(room_data.freq_bins_rad, device=device)
        r = 1.0 if new_sampling_radius in (1.0, None) else float(new_sampling_radius)
        self.z_values = torch.polar(r * torch.ones_like(w), w)
        self.rir_mag_response = room_data.rir_mag_response
        self.late_rir_mag_response = room_data.late_rir_mag_response
        self.early_rir_mag_response = room_data.early_rir_mag_response
        self.early_rir_time = getattr(room_data, 'early_rir_time', None)       # (R, mixing time) float64, or None
        self.edr_store = None     # (T_db (R,frames,F), sum_abs (R,)) once precomputed
        self.edc_store = None     # ((start, length), T_db (R,length))

    def __len__(self):
        return self.listener_positions.shape[0]

    def __getitem__(self, idx: int):
        return int(idx)

    def early_response_c128(self, idx) -> torch.Tensor:
        """The early responses of the receivers ``idx`` as the reference holds them: complex128 rfft of the float64 early
        RIRs (dataloader.py:250, :300-325) -- from the time-domain store through the float64 transform of csrc/fft64.hip
        (the resident ``early_rir_mag_response`` is its complex64 rounding of a float32 transform: 1e-6 of a row's rms).
        Falls back to that store where the dataset was not built from time signals."""
        K = self.z_values.shape[-1]
        nfft = 2 * (K - 1)
        Et = self.early_rir_time
        if Et is None or nfft < 2 or nfft & (nfft - 1) or Et.shape[1] > nfft:
            return self.early_rir_mag_response[idx].to(torch.complex128)
        idx = torch.as_tensor(idx, device=Et.device).reshape(-1)
        return ops.rfft_pow2_f64(Et[idx].contiguous(), nfft, K)

    # model-independent loss targets for the whole grid (SURVEY §8d "precomputed once")
    def precompute_decay_targets(self, win: int, edc_start: int, edc_len: int, chunk: int = 64):
        if (self.edr_store is not None and self.edc_store is not None and self.edr_store[0] == win
                and self.edc_store[0] == (edc_start, edc_len)):
            return                      # the stores already hold these targets
        R = len(self)
        nf = ops.stft_nframes(self.rir_mag_response.shape[-1], win)
        T_edr = torch.empty((R, nf, win // 2 + 1), dtype=torch.float32, device=self.device)
        sum_abs = torch.empty(R, dtype=torch.float32, device=self.device)
        T_edc = torch.empty((R, edc_len), dtype=torch.float32, device=self.device)
        K = self.rir_mag_response.shape[-1]
        for r0 in range(0, R, chunk):
            x = ops.irfft_odd_fwd(self.rir_mag_response[r0:r0 + chunk], K)
            P = ops.stft_power(x, win)
            t, s = ops.edr_target(P)
            T_edr[r0:r0 + chunk], sum_abs[r0:r0 + chunk] = t, s
            T_edc[r0:r0 + chunk] = ops.edc_target(x, edc_start, edc_len)
        self.edr_store = (win, T_edr, sum_abs)
        self.edc_store = ((edc_start, edc_len), T_edc)

    def collate(self, indices, lean: bool = False) -> Dict:
        """Batch dict with the reference's keys (custom_collate :674-704) + 'receiver_index'.
        ``lean`` skips gathering responses the training step does not read (the late response
        always; the full target response when its EDR / EDC are already in the stores).
        ``lean="rows"`` gathers NOTHING: the per-receiver entries are the dataset-level stores over
        all receivers and 'row_index' holds the batch's indices into them -- only consumers that
        honour 'row_index' (VarReceiverPosTrainer._step_losses and the kernels under it) may be
        given such a batch."""
        if torch.is_tensor(indices):
            idx = indices          # device index tensor (static buffer under graph replay)
        else:
            idx = torch.as_tensor(list(indices), dtype=torch.long, device=self.device)
        B = idx.numel()
        if lean == "rows":
            if self.edr_store is None or self.edc_store is None:
                raise RuntimeError('collate(lean="rows") needs precompute_decay_targets() first')
            return {
                'z_values': self.z_values,
                'source_position': self.source_position[0].expand(B, -1),
                'norm_listener_position': self.norm_listener_position,
                'listener_position': self.listener_positions,
                'target_early_response': self.early_rir_mag_response,
                'edr_target': self.edr_store,
                'edc_target': self.edc_store,
                'receiver_index': idx,
                'row_index': idx,
            }
        batch = {
            'z_values': self.z_values,
            'source_position': self.source_position[0].expand(B, -1),
            'listener_position': self.listener_positions[idx],
            'norm_listener_position': self.norm_listener_position[idx],
            'target_early_response': self.early_rir_mag_response[idx],
            'receiver_index': idx,
        }
        if not lean:
            batch['target_late_response'] = self.late_rir_mag_response[idx]
        if not lean or self.edr_store is None or self.edc_store is None:
            batch['target_rir_response'] = self.rir_mag_response[idx]
        if self.edr_store is not None:
            win, T, s = self.edr_store
            batch['edr_target'] = (win, T[idx], s[idx])
        if self.edc_store is not None:
            (st, ln), T = self.edc_store
            batch['edc_target'] = ((st, ln), T[idx])
        return batch


def split_dataset(dataset, train_valid_split_ratio: float, test_ratio: Optional[float] = None,
                  test_seed: int = 4314) -> Tuple[List[int], List[int], List[int]]:
    """Index lists (train, valid, test): fixed test split drawn with its own generator
    (reference create_fixed_test_split :707-728), then a random split of the remainder with the
    global generator (reference :829-843)."""
    n = len(dataset)
    test: List[int] = []
    remaining = list(range(n))
    if test_ratio:
        gen = torch.Generator().manual_seed(test_seed)
        perm = torch.randperm(n, generator=gen).tolist()
        nt = int(n * test_ratio)
        test, remaining = perm[:nt], perm[nt:]
    nrem = len(remaining)
    ntrain = int(train_valid_split_ratio * nrem)
    perm = torch.randperm(nrem).tolist()
    train = [remaining[i] for i in perm[:ntrain]]
    valid = [remaining[i] for i in perm[ntrain:]]
    return train, valid, test


def rank_shard(chunk: Sequence[int], rank: int, world_size: int) -> List[int]:
    """This rank's share of one (global) batch: EQUAL shares on every rank -- ``len(chunk) // world_size`` items,
    rank r taking chunk[r::world_size] of the first ``world_size * share`` -- so that every rank runs the same
    number of steps with the same collectives (a rank with an empty or a longer shard would leave the others
    waiting in an all-reduce).  Up to ``world_size - 1`` items of a ragged tail are dropped; a chunk smaller than
    the world gives [] on EVERY rank (the caller skips that step everywhere)."""
    share = len(chunk) // world_size
    return list(chunk[:share * world_size][rank::world_size]) if share > 0 else []


class GridLoader:
    """Minimal DataLoader over index lists: shuffles with the global torch generator and yields
    collated batch dicts.  Optionally shards every batch over the ranks of a process group
    (data-parallel over receiver positions): every rank then takes an EQUAL share of every batch
    (``rank_shard``) and batches smaller than the world are skipped on all ranks alike -- the ranks must shuffle with
    the same generator state (seed them identically), as they must for any data-parallel loader."""

    def __init__(self, dataset: MultiRIRDataset, indices: Sequence[int], batch_size: int,
                 shuffle: bool = True, drop_last: bool = False, rank: int = 0, world_size: int = 1,
                 lean: bool = False):
        self.dataset = dataset
        self.indices = list(indices)
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.drop_last = drop_last
        self.rank, self.world_size = rank, world_size
        self.lean = lean

    def _bounds(self):
        n, B = len(self.indices), self.batch_size
        nb = n // B if self.drop_last else (n + B - 1) // B
        out = [(i * B, min((i + 1) * B, n)) for i in range(nb)]
        if self.world_size > 1:
            out = [(a, b) for a, b in out if b - a >= self.world_size]
        return out

    def __len__(self):
        return len(self._bounds())

    def __iter__(self):
        order = self.indices
        if self.shuffle:
            perm = torch.randperm(len(order)).tolist()
            order = [order[i] for i in perm]
        for a, b in self._bounds():
            chunk = order[a:b]
            if self.world_size > 1:
                chunk = rank_shard(chunk, self.rank, self.world_size)
            yield self.dataset.collate(chunk, lean=self.lean)


# ------------------------------------------------------------------------------------------------------------
# Single measured / simulated RIR (config 1: one source-receiver pair) and the reference's loader entry points
# ------------------------------------------------------------------------------------------------------------
class RIRData:
    """One RIR with its full / early / late frequency responses (reference dataloader.py:76-180).  Host-side numpy
    like the reference: one response of a few seconds.  ``early_late_split`` applies the 5 ms fades IN PLACE on
    views of ``rir`` exactly as the reference does (:160-170), so ``rir_mag_response`` -- a property evaluated
    afterwards -- is the spectrum of the faded signal there too."""

    def __init__(self, common_decay_times, band_centre_hz=None, amplitudes=None, room_dims=None,
                 absorption_coeffs=None, mixing_time_ms: float = 20.0, nfft: Optional[int] = None, wav_path=None,
                 rir: Optional[np.ndarray] = None, sample_rate: Optional[float] = None):
        if wav_path is None and rir is None:
            raise AttributeError("Either the path to the wav file or the RIR itself must be specified")
        if wav_path is not None and rir is None:
            assert str(wav_path).endswith('.wav'), "provide the path to the .wav file"
            from scipy.io import wavfile           # (soundfile in the reference)
            try:
                sample_rate, rir = wavfile.read(str(wav_path))
            except Exception as exc:
                raise FileNotFoundError(f"File was not found at {str(wav_path)}") from exc
            if np.issubdtype(rir.dtype, np.integer):
                rir = rir.astype(np.float64) / np.iinfo(rir.dtype).max
            rir = np.array(rir, dtype=np.float64)
        self.rir = rir
        self.sample_rate = sample_rate
        self.common_decay_times = common_decay_times
        self.band_centre_hz = band_centre_hz
        self.amplitudes = amplitudes
        self.mixing_time_ms = mixing_time_ms
        self.room_dims = room_dims
        self.absorption_coeffs = absorption_coeffs
        self.nfft = nfft
        self.early_late_split()

    @property
    def num_freq_bins(self) -> int:
        if self.nfft is not None:
            return self.nfft
        max_rt60_samps = np.asarray(self.common_decay_times).max() * self.sample_rate
        return int(np.power(2, np.ceil(np.log2(max_rt60_samps))))

    @property
    def freq_bins_rad(self) -> np.ndarray:
        return np.fft.rfftfreq(self.num_freq_bins) * 2 * np.pi

    @property
    def freq_bins_hz(self) -> np.ndarray:
        return np.fft.rfftfreq(self.num_freq_bins, d=1.0 / self.sample_rate)

    @property
    def rir_mag_response(self) -> np.ndarray:
        return np.fft.rfft(self.rir, n=self.num_freq_bins)

    def early_late_split(self, win_len_ms: float = 5.0):
        mixing_time_samps = ms_to_samps(self.mixing_time_ms, self.sample_rate)
        win_len_samps = ms_to_samps(win_len_ms, self.sample_rate)
        window = np.hanning(win_len_samps)
        fade_in_win, fade_out_win = window[:win_len_samps // 2], window[win_len_samps // 2:]
        self.early_rir = self.rir[:mixing_time_samps]           # views: the fades land in self.rir
        self.late_rir = self.rir[mixing_time_samps:]
        self.early_rir[-win_len_samps // 2:] *= fade_out_win
        self.late_rir[:win_len_samps // 2] *= fade_in_win
        self.late_rir_mag_response = np.fft.rfft(self.late_rir, n=self.num_freq_bins)
        self.early_rir_mag_response = np.fft.rfft(self.early_rir, n=self.num_freq_bins)


class SingleRIRDataset(torch.utils.data.Dataset):
    """The bins of ONE response as dataset items (reference :603-658): a batch holds a set of bins; with
    ``batch_size = len(dataset)`` and no shuffling it is the whole grid, which is how the single-position models
    are stepped (DiffGFDNSinglePos takes (K,) tensors)."""

    def __init__(self, device, rir_data: RIRData, new_sampling_radius: Optional[float] = None):
        self.device = device
        w = torch.tensor(rir_data.freq_bins_rad, device=device)
        if new_sampling_radius in (1.0, None):
            self.z_values = torch.polar(torch.ones_like(w), w)
        else:
            assert new_sampling_radius > 1.0
            self.z_values = torch.polar(new_sampling_radius * torch.ones_like(w), w)
        self.rir_mag_response = torch.tensor(rir_data.rir_mag_response, device=device)
        self.late_rir_mag_response = torch.tensor(rir_data.late_rir_mag_response, device=device)
        self.early_rir_mag_response = torch.tensor(rir_data.early_rir_mag_response, device=device)

    def __len__(self):
        return len(self.z_values)

    def __getitem__(self, idx: int) -> Dict:
        return {'z_values': self.z_values[idx], 'target_rir_response': self.rir_mag_response[idx],
                'target_early_response': self.early_rir_mag_response[idx],
                'target_late_response': self.late_rir_mag_response[idx]}


def to_device(data_class, device):
    """Move all tensor / ndarray attributes to ``device`` (reference :661-671)."""
    for name, value in list(data_class.__dict__.items()):
        if isinstance(value, torch.Tensor):
            setattr(data_class, name, value.to(device))
        elif isinstance(value, np.ndarray):
            setattr(data_class, name, torch.tensor(value, device=device))
    if hasattr(data_class, 'device'):
        data_class.device = device
    return data_class


def get_device():
    return torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def create_fixed_test_split(dataset, test_ratio: float = 0.1, seed: int = 4314) -> Tuple[List[int], List[int]]:
    """(test indices, remaining indices) drawn with a generator of their own (reference :707-724)."""
    gen = torch.Generator().manual_seed(seed)
    perm = torch.randperm(len(dataset), generator=gen).tolist()
    nt = int(len(dataset) * test_ratio)
    return perm[:nt], perm[nt:]


def get_dataloader(dataset, batch_size: int, shuffle: bool = True, device='cpu', drop_last: bool = True,
                   custom_collate_fn=None, indices: Optional[Sequence[int]] = None):
    """Batches of ``dataset`` (reference :748-772).  A grid dataset yields collated batch dicts straight from its
    device-resident stores (:class:`GridLoader`: an index list per batch, no per-item gather on the host); any
    other dataset goes through ``torch.utils.data.DataLoader`` as in the reference."""
    if isinstance(dataset, MultiRIRDataset):
        idx = list(range(len(dataset))) if indices is None else list(indices)
        return GridLoader(dataset, idx, batch_size, shuffle=shuffle, drop_last=drop_last)
    kw = dict(batch_size=batch_size, shuffle=shuffle, drop_last=drop_last,
              generator=torch.Generator(device=device) if shuffle else None)
    if custom_collate_fn is not None:
        kw['collate_fn'] = custom_collate_fn
    return torch.utils.data.DataLoader(dataset, **kw)


def load_dataset(room_data, device, train_valid_split_ratio: float = 0.8, batch_size: int = 32, shuffle: bool = True,
                 new_sampling_radius: Optional[float] = None, drop_last: bool = False,
                 hold_out_test_set: bool = False, test_set_ratio: Optional[float] = None,
                 test_set_seed: Optional[int] = None):
    """Training / validation (/ test) loaders of a :class:`RoomDataset`, or the single loader of an
    :class:`RIRData` (reference :780-867; same arguments)."""
    if isinstance(room_data, RoomDataset):
        dataset = MultiRIRDataset(device, room_data, new_sampling_radius=new_sampling_radius)
        if hold_out_test_set:
            train, valid, test = split_dataset(dataset, train_valid_split_ratio,
                                               test_ratio=0.1 if test_set_ratio is None else test_set_ratio,
                                               **({} if test_set_seed is None else {'test_seed': test_set_seed}))
        else:
            train, valid, test = split_dataset(dataset, train_valid_split_ratio)
        loaders = [GridLoader(dataset, train, batch_size, shuffle=shuffle, drop_last=drop_last),
                   GridLoader(dataset, valid, batch_size, shuffle=shuffle, drop_last=drop_last)]
        if hold_out_test_set:
            loaders.append(GridLoader(dataset, test, batch_size, shuffle=False, drop_last=False))
        return tuple(loaders)
    if isinstance(room_data, RIRData):
        dataset = SingleRIRDataset(device, room_data, new_sampling_radius=new_sampling_radius)
        return get_dataloader(dataset, batch_size=batch_size, shuffle=shuffle, device=device, drop_last=drop_last)
    raise TypeError("load_dataset: RoomDataset or RIRData expected")
