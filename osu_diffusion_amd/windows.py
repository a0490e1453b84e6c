"""The training-window contract of the reference loader (data_loading.py:139-203, 206-377), minus `.osu` parsing.

A *sequence* is the reference's hit-object tensor (19, L): rows 0-1 position in osu! pixels (512 x 384 playfield),
row 2 time in ms, rows 3..18 a one-hot of the 16 hit-object types (data_loading.py:32-40, 128-135).  Everything
from that tensor to the `((x, o, c), y)` batches the DiT path consumes is restated here:

* `calc_distances`, `random_flip`, `split_and_process_sequence[_no_augment]`, `window_and_relative_time` — the
  per-sequence / per-window tensor maths, consuming Python's `random` in the reference's order, so a run seeded
  like the reference yields identical windows (tests/golden/g9_windows.npz pins this);
* `WindowIterable` / `InterleavingIterable` — overlapping windows of `seq_len` every `stride` from a random phase,
  one source after the other, optionally `cycle_length` sources interleaved round-robin
  (BeatmapDatasetIterable / InterleavingBeatmapDatasetIterable);
* `WindowDataset`, `worker_init_fn`, `get_data_loader` — the torch `IterableDataset` with the reference's
  DataLoader-worker split; the per-rank split of train.py:165-170 is `training.shard_range`;
* `cache_dataset`, `CachedDataset`, `get_cached_data_loader` — the reference's cached-epoch variant (:414-475).

Reading `.osu` files needs the third-party `slider` package (absent here, unpinned upstream): a *source* is therefore
anything `open_fn` turns into a (19, L) tensor — by default a `(name, tensor)` pair, with the class label taken from the
first six characters of the name exactly as the reference does with beatmap file names.
"""
import math
import os
import random
from typing import Callable, Iterable, Optional, Sequence

import torch
from torch.utils.data import DataLoader, Dataset, IterableDataset

from .positional_embedding import timestep_embedding

playfield_size = torch.tensor((512, 384))
feature_size = 19


def calc_distances(seq: torch.Tensor) -> torch.Tensor:
    """Distance of every hit object to its predecessor (the first one to the playfield centre) — data_loading.py:146-151."""
    prev = torch.roll(seq[:2, :], 1, 1)
    prev[0, 0] = 256
    prev[1, 0] = 192
    return torch.linalg.vector_norm(seq[:2, :] - prev, ord=2, dim=0)


def random_flip(xy: torch.Tensor) -> torch.Tensor:
    """In-place horizontal / vertical mirror, each with probability 1/2 (two `random.random()` draws) — :139-144."""
    if random.random() < 0.5:
        xy[0] = 512 - xy[0]
    if random.random() < 0.5:
        xy[1] = 384 - xy[1]
    return xy


def _process(seq: torch.Tensor, augment: bool):
    dist = calc_distances(seq)  # before the flip, as the reference does (distances are flip invariant anyway)
    xy = random_flip(seq[:2, :]) if augment else seq[:2, :]
    x = xy / playfield_size.to(seq.device).unsqueeze(1)
    o = seq[2, :]
    c = torch.concatenate([timestep_embedding(dist, 128).T, seq[3:, :]], 0)
    return (x, o, c), seq.shape[1]


def split_and_process_sequence(seq: torch.Tensor):
    """(19, L) -> ((x (2,L) in [0,1], o (L) ms, c (144,L)), L) with the random-flip augmentation — :154-169."""
    return _process(seq, True)


def split_and_process_sequence_no_augment(seq: torch.Tensor):
    """Same without augmentation (sampling path) — :172-187."""
    return _process(seq, False)


def window_and_relative_time(seq, s: int, e: int):
    """Window [s, e) with times made relative to its first object plus a random offset in [0, 1e5) ms — :195-203."""
    seq_x, seq_o, seq_c = seq
    return seq_x[:, s:e], seq_o[s:e] - seq_o[s] + random.random() * 100000, seq_c[:, s:e]


def _open_pair(source):
    return source[1]


def label_of(source) -> int:
    """Class label = the beatmap id in the first six characters of the file / source name — :255."""
    name = source[0] if isinstance(source, (tuple, list)) else source
    return int(os.path.basename(str(name))[:6])


class WindowIterable:
    """Overlapping windows over a list of sources, one source after the other (BeatmapDatasetIterable, :206-267).

    A source shorter than `seq_len` (after its random phase) yields nothing.  `seq_func(opened)` returns
    `(processed_sequence, length)`; `win_func(processed, s, e)` cuts one window.
    """

    def __init__(self, sources: Sequence, seq_len: int, stride: int, seq_func: Callable = split_and_process_sequence,
                 win_func: Callable = window_and_relative_time, open_fn: Callable = _open_pair,
                 label_fn: Callable = label_of):
        self.sources = sources
        self.seq_len, self.stride = seq_len, stride
        self.seq_func, self.win_func, self.open_fn, self.label_fn = seq_func, win_func, open_fn, label_fn
        self.index = 0
        self.current_idx = 0
        self.current_seq = None
        self.current_seq_len = -1
        self.seq_index = 0

    def __iter__(self):
        return self

    def __next__(self):
        while self.current_seq is None or self.seq_index + self.seq_len > self.current_seq_len:
            if self.index >= len(self.sources):
                raise StopIteration
            source = self.sources[self.index]
            self.current_idx = self.label_fn(source)
            self.current_seq, self.current_seq_len = self.seq_func(self.open_fn(source))
            self.seq_index = random.randint(0, self.stride - 1)
            self.index += 1
        window = self.win_func(self.current_seq, self.seq_index, self.seq_index + self.seq_len)
        self.seq_index += self.stride
        return window, self.current_idx


class InterleavingIterable:
    """`cycle_length` sub-iterables over contiguous slices of the sources, served round-robin (:270-304)."""

    def __init__(self, sources: Sequence, iterable_factory: Callable, cycle_length: int):
        per = int(math.ceil(len(sources) / float(cycle_length)))
        self.workers = [iterable_factory(sources[i * per: min(len(sources), (i + 1) * per)]) for i in range(cycle_length)]
        self.cycle_length = cycle_length
        self.index = 0

    def __iter__(self):
        return self

    def __next__(self):
        for _ in range(len(self.workers)):
            try:
                self.index = self.index % len(self.workers)
                item = next(self.workers[self.index])
                self.index += 1
                return item
            except StopIteration:
                self.workers.remove(self.workers[self.index])
        raise StopIteration


class WindowIterableFactory:
    """Picklable `sources -> WindowIterable` (BeatmapDatasetIterableFactory, :394-411)."""

    def __init__(self, seq_len: int, stride: int, seq_func: Callable = split_and_process_sequence,
                 win_func: Callable = window_and_relative_time, open_fn: Callable = _open_pair,
                 label_fn: Callable = label_of):
        self.args = (seq_len, stride, seq_func, win_func, open_fn, label_fn)

    def __call__(self, sources):
        return WindowIterable(sources, *self.args)


class WindowDataset(IterableDataset):
    """Sources `[start, end)` of a catalogue as a stream of `((x, o, c), y)` windows (BeatmapDataset, :307-362).

    `catalogue(start, end)` lists the sources of that index range (the reference lists `TrackNNNNN/beatmaps/*`);
    a plain list is sliced.  DataLoader workers split `[start, end)` among themselves through `worker_init_fn`.
    """

    def __init__(self, catalogue, start: int, end: int, iterable_factory: Callable, cycle_length: int = 1,
                 shuffle: bool = False):
        super().__init__()
        self.catalogue, self.start, self.end = catalogue, start, end
        self.iterable_factory, self.cycle_length, self.shuffle = iterable_factory, cycle_length, shuffle

    def _sources(self):
        if callable(self.catalogue):
            return list(self.catalogue(self.start, self.end))
        return list(self.catalogue[self.start:self.end])

    def __iter__(self):
        sources = self._sources()
        if self.shuffle:
            random.shuffle(sources)
        if self.cycle_length > 1:
            return InterleavingIterable(sources, self.iterable_factory, self.cycle_length)
        return self.iterable_factory(sources)


def worker_init_fn(worker_id: int) -> None:
    """Give DataLoader worker `worker_id` its contiguous share of the dataset's index range (:366-376)."""
    info = torch.utils.data.get_worker_info()
    ds = info.dataset
    per = int(math.ceil((ds.end - ds.start) / float(info.num_workers)))
    ds.start = ds.start + worker_id * per
    ds.end = min(ds.start + per, ds.end)


def get_data_loader(catalogue, start: int, end: int, iterable_factory: Callable, cycle_length: int = 1,
                    batch_size: int = 1, num_workers: int = 0, shuffle: bool = False, pin_memory: bool = False,
                    drop_last: bool = False) -> DataLoader:
    """DataLoader over a WindowDataset; default collate stacks to ((B,2,T), (B,T), (B,144,T)), (B,) — :478-512."""
    ds = WindowDataset(catalogue, start, end, iterable_factory, cycle_length, shuffle)
    return DataLoader(ds, batch_size=batch_size, worker_init_fn=worker_init_fn, num_workers=num_workers,
                      pin_memory=pin_memory, drop_last=drop_last, persistent_workers=num_workers > 0)


class CachedDataset(Dataset):
    """Map-style dataset over windows produced once and kept in memory (data_loading.py:414-424)."""

    def __init__(self, cached_data):
        self.cached_data = cached_data

    def __getitem__(self, index):
        return self.cached_data[index]

    def __len__(self):
        return len(self.cached_data)


def cache_dataset(out_path: str, catalogue, start: int, end: int, iterable_factory: Callable, cycle_length: int = 1) -> int:
    """Run the window stream once (no shuffling) and save every `((x, o, c), y)` item to `out_path` (:427-452); returns the
    number of windows.  Parsing `.osu` files costs tens of ms per map, so a cached epoch keeps the GPU fed without workers."""
    items = list(WindowDataset(catalogue, start, end, iterable_factory, cycle_length, shuffle=False))
    torch.save(items, out_path)
    return len(items)


def get_cached_data_loader(data_path: str, batch_size: int = 1, num_workers: int = 0, shuffle: bool = False,
                           pin_memory: bool = False, drop_last: bool = False) -> DataLoader:
    """DataLoader over a file written by `cache_dataset` (:455-475)."""
    return DataLoader(CachedDataset(torch.load(data_path, weights_only=False)), batch_size=batch_size, num_workers=num_workers,
                      pin_memory=pin_memory, drop_last=drop_last, persistent_workers=num_workers > 0, shuffle=shuffle)


def synthetic_sequences(n: int, min_len: int = 96, max_len: int = 600, seed: int = 0, first_id: int = 100000):
    """`n` seeded stand-ins for parsed beatmaps: `(name, (19, L) tensor)` with the statistics of real maps (positions inside
    the playfield, 50..600 ms gaps, one type per object); the name starts with a six-digit id = the class label."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        L = int(torch.randint(min_len, max_len + 1, (1,), generator=g))
        seq = torch.zeros(feature_size, L)
        seq[0] = torch.rand(L, generator=g) * 512
        seq[1] = torch.rand(L, generator=g) * 384
        seq[2] = torch.cumsum(torch.randint(50, 601, (L,), generator=g).float(), 0)
        types = torch.randint(0, 16, (L,), generator=g)
        seq[3 + types, torch.arange(L)] = 1.0
        out.append((f"{first_id + i:06d} synthetic map {i}.osu", seq))
    return out
