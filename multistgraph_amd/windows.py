"""Multi-temporal-head windows kept on the device.

The reference materialises every sample's input as a (96, N, F) window (``MTHDataset._generate_input_data``,
reference libcity/data/dataset/dataset_subclass/mth_dataset.py:31-160): closeness = the ``len_closeness``
blocks of ``input_window`` steps right before the label, period / trend = one block ``interval * 24 h`` before
the label each, stacked as [closeness oldest -> newest | period | trend] - ~1-2 GB of float64 host arrays that
are deep-copied per batch and shipped over PCIe (data/utils.py:68-72, batch.py:43-57).

Here the raw series (T, N, F) stays resident in HBM and a batch is just B label-start indices: the head-fusion
prologue of the hot path gathers the window rows itself (matgcn_forward_series), so no window is ever built.
``window_offsets`` restates the reference's index arithmetic; it is pinned against the reference's own functions
by tests/golden/windows_small.npz.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def window_offsets(input_window: int = 24, len_closeness: int = 2, len_period: int = 1, len_trend: int = 1,
                   interval_period: int = 7, interval_trend: int = 28, points_per_hour: int = 1,
                   hour_each_day: int = 24) -> np.ndarray:
    """Offsets (relative to the label start) of the x_steps rows of one sample, in the reference's order
    (mth_dataset.py:31-60 ``_search_data``, :82-104, :145-158).  Each block is ``input_window`` consecutive rows."""
    def blocks(count: int, units: float) -> List[int]:
        starts = [-int(points_per_hour * units * i) for i in range(1, count + 1)]
        return starts[::-1]                                   # oldest first (:60)

    rel: List[int] = []
    for start in blocks(len_closeness, input_window / points_per_hour):
        rel += list(range(start, start + input_window))
    for start in blocks(len_period, interval_period * hour_each_day):
        rel += list(range(start, start + input_window))
    for start in blocks(len_trend, interval_trend * hour_each_day):
        rel += list(range(start, start + input_window))
    return np.asarray(rel, dtype=np.int32)


def valid_label_starts(series_steps: int, rel: np.ndarray, input_window: int = 24) -> np.ndarray:
    """Label starts the reference keeps: every block inside the series and ``input_window`` target rows available
    (mth_dataset.py:48-57, :79-80)."""
    lo = int(-rel.min())
    hi = series_steps - input_window
    return np.arange(lo, hi + 1, dtype=np.int32) if hi >= lo else np.zeros(0, dtype=np.int32)


def check_label_starts(label_starts, rel, output_window: int, series_steps: int) -> None:
    """The range contract of the series entry points (include/matgcn.h), checked on the host where a table of label
    starts is built: every window row label_start + rel[s] and every target row label_start + o, o < output_window,
    must lie inside the series.  Raises ValueError naming the offending extreme.  (The kernels clamp and count a
    violation instead of faulting, matgcn_series_violations; this is the loud, early form.)"""
    ls = np.asarray(label_starts).reshape(-1)
    if ls.size == 0:
        return
    rel = np.asarray(rel).reshape(-1)
    lo, hi = int(ls.min()), int(ls.max())
    first = lo + (int(rel.min()) if rel.size else 0)
    if first < 0 or lo < 0:
        raise ValueError("label start %d reaches row %d: before the series (earliest window offset %d)" % (
            lo, first, int(rel.min()) if rel.size else 0))
    last = hi + max(int(output_window) - 1, int(rel.max()) if rel.size else 0)
    if last >= int(series_steps):
        raise ValueError("label start %d reaches row %d: the series has %d rows (%d target rows per sample)" % (
            hi, last, int(series_steps), int(output_window)))


def gather_windows(series: np.ndarray, starts: np.ndarray, rel: np.ndarray, output_window: int) -> Tuple[np.ndarray, np.ndarray]:
    """Host restatement of ``_generate_input_data`` for the given label starts: (B, x_steps, N, F) sources and
    (B, output_window, N, F) targets.  Test infrastructure / cross-check only - the product path never builds X."""
    idx = starts[:, None].astype(np.int64) + rel[None, :].astype(np.int64)
    x = series[idx]
    y = series[starts[:, None].astype(np.int64) + np.arange(output_window)[None, :]]
    return x, y


def split_samples(num_samples: int, train_rate: float = 0.7, eval_rate: float = 0.15) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Sample indices of the train / validation / test parts exactly as TrafficStateDataset._split_train_val_test cuts
    them (libcity/data/dataset/traffic_state_datatset.py:823-834): python round() (banker's rounding) of the test and
    train shares, validation = the rest, test = the LAST num_test samples - written ``x[-num_test:]`` in the
    reference, which is the WHOLE array when num_test rounds to 0; restated as written."""
    test_rate = 1 - train_rate - eval_rate
    num_test = round(num_samples * test_rate)
    num_train = round(num_samples * train_rate)
    num_val = num_samples - num_test - num_train
    idx = np.arange(num_samples, dtype=np.int64)
    return idx[:num_train], idx[num_train:num_train + num_val], idx[-num_test:]


def pad_with_last_sample(indices: np.ndarray, batch_size: int) -> np.ndarray:
    """Repeat the last sample until the part is a whole number of batches (libcity/data/utils.py:53-61; MTHDataset.json
    sets pad_with_last_sample = true), so that every batch has exactly batch_size samples."""
    indices = np.asarray(indices)
    num_padding = (batch_size - (len(indices) % batch_size)) % batch_size
    return np.concatenate([indices, np.repeat(indices[-1:], num_padding, axis=0)], axis=0)


def epoch_batches(label_starts: np.ndarray, part: np.ndarray, batch_size: int, shuffle: bool = False,
                  rng: np.random.Generator = None, rel: np.ndarray = None, output_window: int = None,
                  series_steps: int = None) -> np.ndarray:
    """(batches, batch_size) int32 label starts of one epoch over a part of the samples: the part padded with its
    last sample (the reference pads BEFORE the DataLoader shuffles, data/utils.py:53-74), optionally permuted, cut
    into batches.  A batch on the device is just one row of this table (matgcn_forward_series).  With ``rel``,
    ``output_window`` and ``series_steps`` the table is validated against the series range here, once
    (check_label_starts)."""
    if series_steps is not None:
        check_label_starts(np.asarray(label_starts)[np.asarray(part)], rel if rel is not None else np.zeros(0, np.int32),
                           output_window or 0, series_steps)
    padded = pad_with_last_sample(np.asarray(part), batch_size)
    if shuffle:
        padded = padded[(rng or np.random.default_rng()).permutation(len(padded))]
    return np.asarray(label_starts)[padded].astype(np.int32).reshape(-1, batch_size)
