"""The evaluator's metric table: TrafficStateEvaluator.collect in both modes with the ten
metrics of the shipped TrafficStateEvaluator.json, and the group-std re-transform table of
TrafficStateExecutor.evaluate - against tables the reference's own evaluator / loss functions produced
(tests/golden/metrics_small.npz, generator make_metrics_golden.py).

CPU: the oracle's restatement vs the fixture.  GPU: matgcn_metric_sums / matgcn_metric_table through the C ABI
(multistgraph_amd.evaluator) vs the fixture and vs the oracle - no prediction ever leaves the device.
Tolerance 2e-5 relative per table entry (the reference reduces in fp32, the device in fp64)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR
from oracle import matgcn_oracle as O

GOLD = np.load(os.path.join(GOLDEN_DIR, "metrics_small.npz"))
TOL = 2e-5


def _close(got, want, tol=TOL):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    both_inf = np.isinf(got) & np.isinf(want) & (np.sign(got) == np.sign(want))
    ok = both_inf | (np.abs(got - want) <= tol * np.abs(want) + 1e-9)
    return bool(ok.all()), np.argwhere(~ok)[:5].tolist()


def _table_from_dict(d, out):
    return np.array([[d["%s@%d" % (m, i + 1)] for m in O.EVAL_METRICS] for i in range(out)], dtype=np.float64)


def _descaled(tag):
    p, y = torch.from_numpy(GOLD[tag + "_pred"]), torch.from_numpy(GOLD[tag + "_true"])
    if tag + "_std" in GOLD:
        sd, mu = float(GOLD[tag + "_std"]), float(GOLD[tag + "_mean"])
        p, y = p * np.float32(sd) + np.float32(mu), y * np.float32(sd) + np.float32(mu)
    return p, y


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("mode", ["single", "average"])
def test_oracle_evaluator_table_matches_the_reference(tag, mode):
    p, y = _descaled(tag)
    got = _table_from_dict(O.evaluator_table(p, y, mode), p.shape[1])
    ok, where = _close(got, GOLD["%s_%s" % (tag, mode)])
    assert ok, where
    if tag == "a":     # exact-zero labels: the reference's unmasked MAPE is inf there, the masked one finite
        assert np.isinf(got[:, 1]).any() and np.isfinite(got[:, 5]).all()


def test_oracle_groupstd_table_matches_the_reference():
    p, y = torch.from_numpy(GOLD["c_pred"]), torch.from_numpy(GOLD["c_true"])
    cols = O.groupstd_table(p, y, GOLD["c_all_m"], GOLD["c_all_std"])
    got = np.stack([cols[k] for k in ("MAE", "MSE", "RMSE", "R2", "EVAR", "MAPE")], 1)
    ok, where = _close(got, GOLD["c_table"])
    assert ok, where


def test_device_evaluator_rejects_what_the_reference_rejects():
    from multistgraph_amd.evaluator import DeviceEvaluator
    with pytest.raises(ValueError):
        DeviceEvaluator({"metrics": ["MAE", "F1"]})
    with pytest.raises(TypeError):
        DeviceEvaluator({"metrics": "MAE"})
    with pytest.raises(ValueError):
        DeviceEvaluator({"metrics": ["MAE"], "evaluator_mode": "median"})
    ev = DeviceEvaluator({"metrics": ["MAE"]})
    with pytest.raises(TypeError):
        ev.collect([1, 2])
    with pytest.raises(ValueError):
        ev.collect({"y_true": torch.zeros(2, 3, 4, 1), "y_pred": torch.zeros(2, 3, 5, 1)})


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("mode", ["single", "average"])
def test_device_evaluator_matches_the_reference(tag, mode, lib_built):
    """collect() on de-scaled device tensors (the executor's hand-over), and collect_scaled() with the scaler's
    affine applied in the kernel: the reference's table, entry by entry"""
    from multistgraph_amd.evaluator import ALLOWED_METRICS, DeviceEvaluator
    dev = torch.device("cuda:0")
    p, y = _descaled(tag)
    ev = DeviceEvaluator({"metrics": list(ALLOWED_METRICS), "evaluator_mode": mode})
    ev.collect({"y_true": y.to(dev), "y_pred": p.to(dev)})
    res = ev.evaluate()
    got = _table_from_dict(res, p.shape[1])
    ok, where = _close(got, GOLD["%s_%s" % (tag, mode)])
    assert ok, where
    if tag == "a":     # the same from scaled values: de-scale inside the kernel
        ev2 = DeviceEvaluator({"metrics": list(ALLOWED_METRICS), "evaluator_mode": mode})
        ev2.collect_scaled(torch.from_numpy(GOLD["a_pred"]).to(dev), torch.from_numpy(GOLD["a_true"]).to(dev), 0,
                           float(GOLD["a_mean"]), float(GOLD["a_std"]))
        ok, where = _close(_table_from_dict(ev2.evaluate(), p.shape[1]), GOLD["a_%s" % mode])
        assert ok, where


@pytest.mark.gpu
def test_device_evaluator_batches_streaming_and_series(lib_built):
    """(i) several collect() calls = the mean of the batches' metrics, as TrafficStateEvaluator.evaluate averages them;
    (ii) streaming=True = ONE collect over the concatenation (how the executor evaluates a test set) fed batch by batch;
    (iii) labels gathered from the device-resident series (label_start) give the same table as materialised labels"""
    from multistgraph_amd.evaluator import ALLOWED_METRICS, DeviceEvaluator
    dev = torch.device("cuda:0")
    p, y = _descaled("b")
    cfg = {"metrics": list(ALLOWED_METRICS), "evaluator_mode": "average"}
    parts = [(p[:2], y[:2]), (p[2:], y[2:])]
    ev = DeviceEvaluator(cfg)
    for pp, yy in parts:
        ev.collect({"y_true": yy.to(dev), "y_pred": pp.to(dev)})
    want = np.mean([_table_from_dict(O.evaluator_table(pp, yy, "average"), p.shape[1]) for pp, yy in parts], 0)
    ok, where = _close(_table_from_dict(ev.evaluate(), p.shape[1]), want)
    assert ok, where
    st = DeviceEvaluator(cfg, streaming=True)
    for pp, yy in parts:
        st.collect({"y_true": yy.to(dev), "y_pred": pp.to(dev)})
    ok, where = _close(_table_from_dict(st.evaluate(), p.shape[1]), GOLD["b_average"])
    assert ok, where
    st.clear()
    with pytest.raises(RuntimeError):
        st.table()
    # labels from the raw series: sample b's targets are series[label_start[b] + o]
    rng = np.random.default_rng(2)
    series = rng.standard_normal((200, 61, 3)).astype(np.float32)
    starts = np.array([5, 60, 60, 170, 33], dtype=np.int32)
    out = p.shape[1]
    labels = np.stack([series[s:s + out] for s in starts], 0)            # (B, out, N, F)
    a = DeviceEvaluator(cfg)
    a.collect_scaled(p.to(dev), torch.from_numpy(labels).to(dev), 1, 2.0, 3.0)
    b = DeviceEvaluator(cfg)
    b.collect_scaled(p.to(dev), torch.from_numpy(series).to(dev), 1, 2.0, 3.0, label_start=torch.from_numpy(starts).to(dev))
    assert torch.equal(a.table(), b.table())
    want = O.evaluator_table(p * 3.0 + 2.0, torch.from_numpy(labels[..., 1:3]) * 3.0 + 2.0, "average")
    ok, where = _close(_table_from_dict(a.evaluate(), out), _table_from_dict(want, out))
    assert ok, where


@pytest.mark.gpu
def test_groupstd_table_on_the_device(lib_built):
    """the re-transform table of traffic_state_executor.py:293-322 (per-node affine, clamp at 0, truth_t > 10, R2 / EVAR
    with prediction first) from one pass on the device, whole and fed in two batches"""
    from multistgraph_amd.evaluator import groupstd_table
    dev = torch.device("cuda:0")
    p, y = torch.from_numpy(GOLD["c_pred"]).to(dev), torch.from_numpy(GOLD["c_true"]).to(dev)
    cols, _ = groupstd_table(p, y, GOLD["c_all_m"], GOLD["c_all_std"])
    got = np.stack([cols[k].numpy() for k in ("MAE", "MSE", "RMSE", "R2", "EVAR", "MAPE")], 1)
    ok, where = _close(got, GOLD["c_table"])
    assert ok, where
    _, sums = groupstd_table(p[:4], y[:4], GOLD["c_all_m"], GOLD["c_all_std"])
    cols2, _ = groupstd_table(p[4:], y[4:], GOLD["c_all_m"], GOLD["c_all_std"], sums=sums)
    got2 = np.stack([cols2[k].numpy() for k in ("MAE", "MSE", "RMSE", "R2", "EVAR", "MAPE")], 1)
    ok, where = _close(got2, GOLD["c_table"])
    assert ok, where
    # a scaler in front of the re-transform (the executor de-scales first, :268-273): scalar affine then per node
    cols3, _ = groupstd_table((p - 0.5) / 2.0, (y - 0.5) / 2.0, GOLD["c_all_m"], GOLD["c_all_std"], mean=0.5, std=2.0)
    got3 = np.stack([cols3[k].numpy() for k in ("MAE", "MSE", "RMSE", "R2", "EVAR", "MAPE")], 1)
    ok, where = _close(got3, GOLD["c_table"], tol=2e-4)
    assert ok, where
