#!/usr/bin/env python3
"""Golden vector for the multi-temporal-head windowing, produced by THE REFERENCE'S OWN functions.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_windows_golden.py

MTHDataset cannot be constructed here (no raw data, np.float in its base class), so its two windowing methods
(libcity/data/dataset/dataset_subclass/mth_dataset.py:31-160) are called unbound on a stub that carries exactly
the attributes they read.  Stored: the synthetic series spec, the label starts the reference kept, and checksums
+ a few rows of the (samples, 96, N, F) sources / targets it produced - data only.
"""
import logging
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
from libcity.data.dataset.dataset_subclass.mth_dataset import MTHDataset  # reference


def main():
    cfgs = {
        "user": dict(len_closeness=2, len_period=1, len_trend=1, interval_period=7, interval_trend=28),   # config_user.json
        "dataset_default": dict(len_closeness=2, len_period=1, len_trend=1, interval_period=1, interval_trend=7),
    }
    out = {}
    for tag, c in cfgs.items():
        steps = 24 * 28 + 24 * 2 + 5 if tag == "user" else 24 * 7 + 60
        n, f = 3, 2
        rng = np.random.default_rng(42)
        series = rng.standard_normal((steps, n, f)).astype(np.float32)
        stub = types.SimpleNamespace(points_per_hour=1, hour_each_day=24, input_window=24, output_window=12,
                                     _logger=logging.getLogger("golden"), **c)
        stub._search_data = types.MethodType(MTHDataset._search_data, stub)
        stub._get_sample_indices = types.MethodType(MTHDataset._get_sample_indices, stub)
        src, tgt = MTHDataset._generate_input_data(stub, series)
        # recover the label starts the reference kept: the target block starts at series[start]
        starts = [i for i in range(steps) if stub._get_sample_indices(series, i)[3] is not None]
        assert len(starts) == src.shape[0]
        out[tag + "_steps"] = np.int64(steps)
        out[tag + "_starts"] = np.asarray(starts, dtype=np.int32)
        out[tag + "_src_shape"] = np.asarray(src.shape)
        out[tag + "_src_sum"] = np.float64(src.astype(np.float64).sum())
        out[tag + "_src_first"] = src[0].astype(np.float32)
        out[tag + "_src_last"] = src[-1].astype(np.float32)
        out[tag + "_tgt_last"] = tgt[-1].astype(np.float32)
        for k, v in c.items():
            out[tag + "_" + k] = np.int64(v)
        print(tag, "samples", src.shape, "starts", starts[0], "..", starts[-1])
    # train / validation / test split and last-sample padding, by the reference's own functions on index arrays
    # (traffic_state_datatset.py:806-834, data/utils.py:32-79); 40 samples at 0.98 / 0.01 makes num_test round to 0
    from libcity.data.dataset.traffic_state_datatset import TrafficStateDataset
    from libcity.data.utils import generate_dataloader
    split_cases = [(2905, 0.7, 0.15), (3577, 0.7, 0.15), (1000, 0.6, 0.2), (37, 0.7, 0.15), (40, 0.98, 0.01), (10, 0.5, 0.25)]
    out["split_cases"] = np.asarray(split_cases, dtype=np.float64)
    for i, (ns, tr, ev) in enumerate(split_cases):
        stub = types.SimpleNamespace(train_rate=tr, eval_rate=ev, cache_dataset=False, _logger=logging.getLogger("golden"))
        ids = np.arange(ns, dtype=np.int64)[:, None]
        xt, _, xv, _, xs, _ = TrafficStateDataset._split_train_val_test(stub, ids, ids.copy())
        out["split%d_train" % i], out["split%d_val" % i], out["split%d_test" % i] = xt[:, 0], xv[:, 0], xs[:, 0]
        print("split", ns, tr, ev, "->", len(xt), len(xv), len(xs))
    pad_cases = [(2034, 64), (436, 64), (435, 64), (128, 64), (5, 8), (1, 4)]
    out["pad_cases"] = np.asarray(pad_cases, dtype=np.int64)
    for i, (ns, bs) in enumerate(pad_cases):
        ids = np.arange(ns, dtype=np.int64)[:, None]
        dl_train, dl_eval, dl_test = generate_dataloader(ids, ids[: max(1, ns // 3)], ids[: max(1, ns // 2)], {"X": "float"}, bs, 0,
                                                         shuffle=False, pad_with_last_sample=True)
        out["pad%d_train" % i] = np.asarray(dl_train.dataset.data)[:, 0]
        out["pad%d_eval" % i] = np.asarray(dl_eval.dataset.data)[:, 0]
        out["pad%d_test" % i] = np.asarray(dl_test.dataset.data)[:, 0]
    np.savez_compressed(os.path.join(HERE, "windows_small.npz"), **out)


if __name__ == "__main__":
    main()
