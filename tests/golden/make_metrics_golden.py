#!/usr/bin/env python3
"""Metric fixtures produced by THE REFERENCE'S OWN evaluator and loss functions:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_metrics_golden.py

* TrafficStateEvaluator.collect / evaluate (libcity/evaluator/traffic_state_evaluator.py) with the ten metrics of the
  shipped TrafficStateEvaluator.json, in both modes, on de-scaled synthetic predictions / labels that contain exact
  zeros and labels below min_s;
* the group-std re-transform table of TrafficStateExecutor.evaluate (libcity/executor/traffic_state_executor.py:293-322):
  the executor module cannot be imported here (ray / tensorboard are missing), so the generator applies the reference's
  own loss.masked_*_np functions and sklearn's r2_score / explained_variance_score to the values selected the way
  those lines select them (per node x * All_std + All_m, prediction_t < 0 -> 0, truth_t > 10, prediction first in the
  sklearn calls).
Stored: the inputs themselves (small) and the tables - data only.
"""
import os
import sys

import numpy as np
import torch
from sklearn.metrics import explained_variance_score, r2_score

from libcity.evaluator.traffic_state_evaluator import TrafficStateEvaluator
from libcity.model import loss

# sklearn >= 1.4 returns python floats from r2_score / explained_variance_score; the reference calls .item() on them
# (traffic_state_evaluator.py:116,119 - written against an sklearn that returned numpy scalars): same numbers, re-boxed
_r2, _ev = loss.r2_score_torch, loss.explained_variance_score_torch
loss.r2_score_torch = lambda preds, labels: np.float64(_r2(preds, labels))
loss.explained_variance_score_torch = lambda preds, labels: np.float64(_ev(preds, labels))

HERE = os.path.dirname(os.path.abspath(__file__))
METRICS = ["MAE", "MAPE", "MSE", "RMSE", "masked_MAE", "masked_MAPE", "masked_MSE", "masked_RMSE", "R2", "EVAR"]


def make_inputs(seed, b, out, n, mean, std, zeros=True):
    rng = np.random.default_rng(seed)
    y = rng.standard_normal((b, out, n, 1)).astype(np.float32)
    p = (y + 0.3 * rng.standard_normal((b, out, n, 1))).astype(np.float32)
    if zeros:
        y[0, :, :3, 0] = np.float32(-mean / std)                 # de-scales to (almost) exactly 0
        y[1, 2, 5, 0] = np.float32((5e-5 - mean) / std)          # |label| < min_s after de-scaling -> zeroed
        p[1, 4, 7, 0] = y[1, 4, 7, 0]                            # p == l
    return p, y


def evaluator_tables(pred_ds, true_ds):
    res = {}
    for mode in ("single", "average"):
        ev = TrafficStateEvaluator({"metrics": METRICS, "evaluator_mode": mode})
        ev.collect({"y_true": torch.from_numpy(true_ds.copy()), "y_pred": torch.from_numpy(pred_ds.copy())})
        r = ev.evaluate()
        res[mode] = np.array([[r["%s@%d" % (m, i + 1)] for m in METRICS] for i in range(pred_ds.shape[1])], dtype=np.float64)
    return res


def groupstd(pred_ds, true_ds, all_m, all_std, s_small=10):
    pt = pred_ds * all_std.reshape(1, 1, -1, 1) + all_m.reshape(1, 1, -1, 1)
    tt = true_ds * all_std.reshape(1, 1, -1, 1) + all_m.reshape(1, 1, -1, 1)
    pt = np.where(pt < 0, np.float32(0), pt)
    rows = []
    for rr in range(pred_ds.shape[1]):
        keep = tt[:, rr] > s_small
        pr, tr = pt[:, rr][keep], tt[:, rr][keep]
        rows.append([loss.masked_mae_np(pr, tr), loss.masked_mse_np(pr, tr), loss.masked_rmse_np(pr, tr),
                     r2_score(pr, tr), explained_variance_score(pr, tr), loss.masked_mape_np(pr, tr)])
    return np.array(rows, dtype=np.float64)     # columns MAE, MSE, RMSE, R2, EVAR, MAPE


def main():
    out = {}
    mean, std = 14.41, 29.3            # Baltimore's flow statistics (README.md:52-53)
    p, y = make_inputs(3, 6, 12, 37, mean, std)
    out["a_pred"], out["a_true"], out["a_mean"], out["a_std"] = p, y, np.float32(mean), np.float32(std)
    pd_, yd = (p * np.float32(std) + np.float32(mean)), (y * np.float32(std) + np.float32(mean))
    t = evaluator_tables(pd_, yd)
    out["a_single"], out["a_average"] = t["single"], t["average"]
    # a second, larger case without exact zeros (finite MAPE everywhere), two output channels
    rng = np.random.default_rng(11)
    p2 = rng.standard_normal((5, 24, 61, 2)).astype(np.float32)
    y2 = (p2 + 0.5 * rng.standard_normal((5, 24, 61, 2))).astype(np.float32)
    out["b_pred"], out["b_true"] = p2, y2
    t = evaluator_tables(p2.copy(), y2.copy())
    out["b_single"], out["b_average"] = t["single"], t["average"]
    # group-std re-transform: per-tract mean / std like other_data/*_visit_mstd.pkl (All_m, All_std)
    rng = np.random.default_rng(5)
    all_m = rng.uniform(5.0, 60.0, 37).astype(np.float32)
    all_std = rng.uniform(3.0, 40.0, 37).astype(np.float32)
    p3, y3 = make_inputs(9, 6, 12, 37, 0.0, 1.0, zeros=False)
    out["c_pred"], out["c_true"], out["c_all_m"], out["c_all_std"] = p3, y3, all_m, all_std
    out["c_table"] = groupstd(p3, y3, all_m, all_std)
    np.savez_compressed(os.path.join(HERE, "metrics_small.npz"), **out)
    print("a single MAE@1..3", out["a_single"][:3, 0], " MAPE@1", out["a_single"][0, 1])
    print("b average R2@24", out["b_average"][-1, 8], " c table row 0", out["c_table"][0])


if __name__ == "__main__":
    main()
