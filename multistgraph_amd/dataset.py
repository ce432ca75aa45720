"""Device-resident counterpart of the reference's ``MTHDataset``.

The reference materialises every sample as a float64 ``(x_steps, N, F)`` window plus an ``(out, N, F)`` label block
(``MTHDataset._generate_input_data``, libcity/data/dataset/dataset_subclass/mth_dataset.py:110-160), deep-copies
every item in the DataLoader's collate function (libcity/data/utils.py:68-72) and ships each batch over PCIe
(``Batch.to_tensor``, libcity/data/batch.py:43-57).  Here the scaled series ``(T, N, F)`` is uploaded ONCE and a
batch is ``batch_size`` label-start indices: the hot path gathers the window rows and the targets on the device
(``matgcn_forward_series``, ``matgcn_masked_mae`` with ``label_start``).

* ``ResidentSeries`` - the core: sample indices, train / validation / test cut, scaler statistics, last-sample
  padding and shuffling exactly as the reference computes them (each step cites its lines), from the raw series alone;
* ``ResidentBatch`` - what the loaders yield: the ``Batch`` surface the executor uses (``to_tensor(device)``,
  ``batch['y']``, ``batch['X']``; libcity/data/batch.py:5-57) over ``series`` + ``label_start``; ``MultiATGCN``'s
  ``calculate_loss`` / ``predict`` dispatch to the series path when they get one;
* ``MTHDatasetResident`` - the plugin ``libcity.data.utils.get_dataset`` resolves by name (libcity/data/utils.py:10-28):
  a subclass of the reference's ``MTHDataset`` that keeps its file loading (ETL stays the reference's Python) and
  replaces only ``get_data``.  INTEGRATION.md shows the one-line registry change and the config key.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import windows

try:  # inside a LibCity checkout: the reference's own base class and scalers
    from libcity.data.dataset.dataset_subclass.mth_dataset import MTHDataset as _MTHBase  # type: ignore
except Exception:  # standalone (tests, bench): the core below needs no LibCity
    _MTHBase = object
try:
    from libcity.utils.normalization import (LogScaler, MinMax01Scaler, MinMax11Scaler, NoneScaler,  # type: ignore
                                             NormalScaler, StandardScaler)
except Exception:  # same arithmetic as libcity/utils/normalization.py:20-113
    class NoneScaler:
        def transform(self, data):
            return data

        def inverse_transform(self, data):
            return data

    class NormalScaler:
        def __init__(self, maxx):
            self.max = maxx

        def transform(self, data):
            return data / self.max

        def inverse_transform(self, data):
            return data * self.max

    class StandardScaler:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def transform(self, data):
            return (data - self.mean) / self.std

        def inverse_transform(self, data):
            return (data * self.std) + self.mean

    class MinMax01Scaler:
        def __init__(self, maxx, minn):
            self.max, self.min = maxx, minn

        def transform(self, data):
            return (data - self.min) / (self.max - self.min)

        def inverse_transform(self, data):
            return data * (self.max - self.min) + self.min

    class MinMax11Scaler:
        def __init__(self, maxx, minn):
            self.max, self.min = maxx, minn

        def transform(self, data):
            return ((data - self.min) / (self.max - self.min)) * 2. - 1.

        def inverse_transform(self, data):
            return ((data + 1.) / 2.) * (self.max - self.min) + self.min

    class LogScaler:
        def __init__(self, eps=0.999):
            self.eps = eps

        def transform(self, data):
            return np.log(data + self.eps)

        def inverse_transform(self, data):
            return np.exp(data) - self.eps


FEATURE_NAME = {"X": "float", "y": "float"}


class ResidentBatch:
    """One batch of a resident loader.  Quacks like libcity.data.batch.Batch for what TrafficStateExecutor touches:
    ``to_tensor(device)`` (:409, :439, :264 - after the first batch a no-op apart from 4 bytes per sample),
    ``batch['y']`` (:268-270: gathered lazily ON THE DEVICE), ``batch['X']`` (the same, for consumers that want
    windows); and carries what the HIP path consumes instead: ``series`` (T, N, F) float32 resident on the device,
    ``label_start`` (B) int32, ``rel_steps``."""

    def __init__(self, owner: "ResidentSeries", label_start: np.ndarray):
        self.owner = owner
        self.feature_name = dict(FEATURE_NAME)
        self._host_starts = np.ascontiguousarray(label_start, dtype=np.int32)
        self.data: Dict[str, object] = {}
        self.range_checked = True      # the owner validated every label start of its table on the host

    def to_tensor(self, device):
        device = torch.device(device)
        self.data["series"] = self.owner.series_on(device)
        self.data["label_start"] = torch.from_numpy(self._host_starts).to(device)
        self.data["rel_steps"] = self.owner.rel
        self.data.pop("X", None)
        self.data.pop("y", None)

    def to_ndarray(self):
        raise TypeError("a resident batch has no host form: the series lives on the device (use batch['X'] / batch['y'] "
                        "after to_tensor and copy them back if a host copy is really needed)")

    def _rows(self, offsets) -> torch.Tensor:
        if "series" not in self.data:
            raise KeyError("call to_tensor(device) first: the resident batch gathers on the device")
        series, ls = self.data["series"], self.data["label_start"]
        idx = ls.long()[:, None] + torch.as_tensor(np.asarray(offsets), dtype=torch.long, device=ls.device)[None, :]
        return series[idx]

    def __contains__(self, key):
        return key in ("X", "y", "series", "label_start", "rel_steps")

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __getitem__(self, key):
        if key == "y":       # targets series[label_start + o] (mth_dataset.py:105), gathered on the device, cached
            if "y" not in self.data:
                self.data["y"] = self._rows(np.arange(self.owner.output_window))
            return self.data["y"]
        if key == "X":       # the materialised window, for consumers other than the HIP path
            if "X" not in self.data:
                self.data["X"] = self._rows(self.owner.rel)
            return self.data["X"]
        if key in self.data:
            return self.data[key]
        raise KeyError("{} is not in the batch".format(key))

    def __setitem__(self, key, value):
        if key in self:
            self.data[key] = value
            if key in ("label_start", "series", "rel_steps"):   # no longer what the owner validated on the host
                self.range_checked = False
        else:
            raise KeyError("{} is not in the batch".format(key))

    def __len__(self):
        return len(self._host_starts)


class _Positions(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


class ResidentSeries:
    """Everything ``TrafficStateDataset.get_data`` derives from the materialised windows, derived from the raw series
    ``(T, N, F)`` (float64 as ``_load_dyna`` returns it) without building one window:

    * samples = the label starts the reference keeps (mth_dataset.py:48-57, 79-80; windows.valid_label_starts), in
      time order; several data files are several series whose samples are concatenated (:791-801);
    * train / validation / test cut by ``round()`` (traffic_state_datatset.py:823-834; windows.split_samples);
    * the scaler's statistics over ``x_train[..., :output_dim]`` / ``y_train[..., :output_dim]`` (:953-955, :903-929):
      a window row that k training windows contain counts k times - a weighted pass over the series rows in fp64;
    * ``scaler.transform`` on the first ``output_dim`` channels in float64, THEN the float32 cast of ``Batch.to_tensor``
      (:956-961, batch.py:53): the resident float32 series holds bit for bit what the reference's windows hold;
    * loaders: each part padded with its last sample to whole batches, train and validation shuffled by a torch
      DataLoader (same sampler, same consumption of torch's RNG as the reference's), test in order (utils.py:53-82).
    """

    def __init__(self, series, *, input_window: int = 24, output_window: int = 24, len_closeness: int = 2,
                 len_period: int = 1, len_trend: int = 1, interval_period: int = 7, interval_trend: int = 28,
                 points_per_hour: int = 1, hour_each_day: int = 24, train_rate: float = 0.7, eval_rate: float = 0.15,
                 batch_size: int = 64, scaler_type: str = "standard", output_dim: int = 1,
                 pad_with_last_sample: bool = True, shuffle: bool = True):
        files = [np.asarray(s) for s in (series if isinstance(series, (list, tuple)) else [series])]
        if any(f.ndim != 3 for f in files):
            raise ValueError("series must be (T, N, F)")
        if len_closeness + len_period + len_trend <= 0:
            raise ValueError("len_closeness + len_period + len_trend must be positive (mth_dataset.py:15)")
        self.input_window, self.output_window, self.batch_size = int(input_window), int(output_window), int(batch_size)
        self.lens = (int(len_closeness), int(len_period), int(len_trend))
        self.output_dim, self.pad_with_last_sample, self.shuffle = int(output_dim), bool(pad_with_last_sample), bool(shuffle)
        self.rel = windows.window_offsets(self.input_window, len_closeness, len_period, len_trend, interval_period,
                                          interval_trend, points_per_hour, hour_each_day)
        # samples of every file, as offsets into the concatenated series (a window never crosses a file boundary:
        # the starts are the ones valid inside their own file)
        starts, base = [], 0
        for f in files:
            starts.append(windows.valid_label_starts(f.shape[0], self.rel, self.input_window).astype(np.int64) + base)
            base += f.shape[0]
        self.label_starts = np.concatenate(starts)
        if len(self.label_starts) == 0:
            raise ValueError("Parameter len_closeness/len_period/len_trend is too large for the time range of the data! "
                             "(mth_dataset.py:134-137)")
        raw = files[0] if len(files) == 1 else np.concatenate(files, 0)
        self.steps, self.num_nodes, self.feature_dim = (int(v) for v in raw.shape)
        self.ext_dim = self.feature_dim - self.output_dim
        if self.output_window > self.input_window:
            raise ValueError("output_window > input_window: the reference's sample filter only guarantees input_window "
                             "target rows (mth_dataset.py:79-80)")
        windows.check_label_starts(self.label_starts, self.rel, self.output_window, self.steps)
        self.parts = dict(zip(("train", "eval", "test"),
                              windows.split_samples(len(self.label_starts), train_rate, eval_rate)))
        self.scaler = self._fit_scaler(scaler_type, raw)
        scaled = np.array(raw, dtype=np.float64, copy=True)
        scaled[..., :self.output_dim] = self.scaler.transform(scaled[..., :self.output_dim])
        self.series_host = torch.from_numpy(scaled.astype(np.float32))     # the cast Batch.to_tensor makes (batch.py:53)
        self._resident: Dict[torch.device, torch.Tensor] = {}
        self._loaders: Optional[Tuple] = None

    # ---- scaler statistics over the training windows, without the windows ------------------------------------------
    def _row_counts(self, part: np.ndarray, offsets: np.ndarray) -> np.ndarray:
        """how many (sample, offset) pairs of ``part`` land on each series row"""
        rows = (self.label_starts[part][:, None] + np.asarray(offsets, dtype=np.int64)[None, :]).reshape(-1)
        return np.bincount(rows, minlength=self.steps).astype(np.float64)

    def _fit_scaler(self, scaler_type: str, raw: np.ndarray):
        od = self.output_dim
        train = self.parts["train"]
        cx = self._row_counts(train, self.rel)                           # x_train rows, with multiplicity
        cy = self._row_counts(train, np.arange(self.output_window))     # y_train rows
        vals = np.asarray(raw[..., :od], dtype=np.float64).reshape(self.steps, -1)
        if scaler_type == "standard":      # StandardScaler(mean=x_train.mean(), std=x_train.std()) (:911-913)
            total = cx.sum() * vals.shape[1]
            mean = float((cx * vals.sum(1)).sum() / total)
            std = float(np.sqrt((cx * np.square(vals - mean).sum(1)).sum() / total))
            return StandardScaler(mean=mean, std=std)
        if scaler_type in ("normal", "minmax01", "minmax11"):     # extrema over x_train AND y_train (:908-922)
            used = (cx + cy) > 0
            mx, mn = float(vals[used].max()), float(vals[used].min())
            if scaler_type == "normal":
                return NormalScaler(maxx=mx)
            return (MinMax01Scaler if scaler_type == "minmax01" else MinMax11Scaler)(maxx=mx, minn=mn)
        if scaler_type == "log":
            return LogScaler()
        if scaler_type == "none":
            return NoneScaler()
        raise ValueError("Scaler type error!")

    # ---- device residency and loaders -----------------------------------------------------------------------------------
    def series_on(self, device) -> torch.Tensor:
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the resident series lives on the GPU (HIP path only); got device %s" % device)
        t = self._resident.get(device)
        if t is None:
            t = self._resident[device] = self.series_host.to(device).contiguous()
        return t

    def part_table(self, part: str) -> np.ndarray:
        """(batches, batch_size) int32 label starts of a part in loader order WITHOUT shuffling: padded with its last
        sample (utils.py:53-61)"""
        idx = self.parts[part]
        if self.pad_with_last_sample:
            idx = windows.pad_with_last_sample(idx, self.batch_size)
        return self.label_starts[idx].astype(np.int32)

    def _loader(self, part: str, shuffle: bool):
        starts = self.part_table(part)
        return torch.utils.data.DataLoader(dataset=_Positions(len(starts)), batch_size=self.batch_size, num_workers=0,
                                           shuffle=shuffle,
                                           collate_fn=lambda pos: ResidentBatch(self, starts[np.asarray(pos, dtype=np.int64)]))

    def loaders(self):
        """(train, eval, test) loaders: train and eval shuffled, test in order (utils.py:74-82)"""
        if self._loaders is None:
            self._loaders = (self._loader("train", self.shuffle), self._loader("eval", self.shuffle),
                             self._loader("test", False))
        return self._loaders

    def data_feature(self) -> dict:
        """the part of get_data_feature() that comes from the data (mth_dataset.py:162-176)"""
        return {"scaler": self.scaler, "num_nodes": self.num_nodes, "feature_dim": self.feature_dim,
                "output_dim": self.output_dim, "ext_dim": self.ext_dim,
                "len_closeness": self.lens[0] * self.input_window, "len_period": self.lens[1] * self.input_window,
                "len_trend": self.lens[2] * self.input_window, "num_batches": len(self.loaders()[0])}


class MTHDatasetResident(_MTHBase):
    """``dataset_class = "MTHDatasetResident"``: the reference's MTHDataset with ``get_data`` replaced.  File loading
    (.geo / .rel in the constructor, .dyna / .ext through ``_load_dyna`` / ``_add_external_information``, .static /
    .gbst) is the parent's; windows, split, scaler and loaders come from ``ResidentSeries``.  No window cache is read or
    written (``cache_dataset`` is ignored: there is nothing to cache)."""

    def __init__(self, config):
        if _MTHBase is object:
            raise ImportError("MTHDatasetResident is the LibCity plugin: it needs libcity.data.dataset on the path "
                              "(use multistgraph_amd.dataset.ResidentSeries directly outside a LibCity checkout)")
        super().__init__(config)
        self.core: Optional[ResidentSeries] = None

    def _raw_series(self) -> List[np.ndarray]:
        """the (T, N, F) arrays ``_generate_data`` cuts its windows from (traffic_state_datatset.py:779-795)"""
        import os
        files = self.data_files.copy() if isinstance(self.data_files, list) else [self.data_files]
        ext_data = None
        if self.load_external and os.path.exists(self.data_path + self.ext_file + ".ext"):
            ext_data = self._load_ext()
        out = []
        for filename in files:
            df = self._load_dyna(filename)
            if self.load_external:
                df = self._add_external_information(df, ext_data)
            out.append(df)
        return out

    def get_data(self):
        import pandas as pd
        if not self.use_3tu and (self.len_closeness + self.len_period + self.len_trend) > 1:
            raise ValueError("use_3tu = false keeps the first input_window rows of X only (traffic_state_datatset.py:"
                             "949-951), which MultiATGCN cannot consume with more than one head: set use_3tu = true "
                             "(MultiATGCN.json:7)")
        if self.normal_external and self.ext_scaler_type != "none":
            raise NotImplementedError("normal_external with ext_scaler = %r is not built (MTHDataset.json ships "
                                      "normal_external = false)" % self.ext_scaler_type)
        self.core = ResidentSeries(
            self._raw_series(), input_window=self.input_window, output_window=self.output_window,
            len_closeness=self.len_closeness, len_period=self.len_period, len_trend=self.len_trend,
            interval_period=self.interval_period, interval_trend=self.interval_trend,
            points_per_hour=self.points_per_hour, hour_each_day=self.hour_each_day, train_rate=self.train_rate,
            eval_rate=self.eval_rate, batch_size=self.batch_size, scaler_type=self.scaler_type,
            output_dim=self.output_dim, pad_with_last_sample=self.pad_with_last_sample)
        self.feature_dim, self.ext_dim = self.core.feature_dim, self.core.ext_dim
        self.scaler, self.ext_scaler = self.core.scaler, NoneScaler()
        # the side tables of get_data (traffic_state_datatset.py:972-982)
        if self.add_static:
            static = pd.read_csv(self.data_path + self.ext_file + ".static").iloc[:, 1:]
            self.static = np.array(static, dtype=float)
        else:
            self.static = None
        if self.groupstd:
            self.ct_visit_mstd = pd.read_csv(self.data_path + self.ext_file + ".gbst").sort_values(
                by="geo_id").reset_index(drop=True)
        else:
            self.ct_visit_mstd = None
        self.coordinate = pd.read_csv(self.data_path + self.ext_file + ".geo")
        self.train_dataloader, self.eval_dataloader, self.test_dataloader = self.core.loaders()
        self.num_batches = len(self.train_dataloader)
        return self.train_dataloader, self.eval_dataloader, self.test_dataloader
