"""MTHDatasetResident / ResidentSeries: the reference's dataset surface over a
device-resident series.

CPU: (i) the core against brute-force windows (scaler statistics with multiplicity, split, padding, loader tables);
(ii) the Batch contract the executor relies on; (iii) - build container only - the reference's OWN MTHDataset on a
synthetic set of atomic files against MTHDatasetResident on the same files: scaler, data_feature, and every window /
label of the test loader and of the (same-seed shuffled) train loader bit for bit.
GPU: one epoch over the resident loaders through the plugin class = the window path vs the oracle."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from helpers import Case, max_norm_err
from multistgraph_amd import windows as W

REFERENCE_ROOT = "/root/reference"
KW = dict(input_window=24, output_window=12, len_closeness=2, len_period=1, len_trend=1, interval_period=2,
          interval_trend=5, train_rate=0.7, eval_rate=0.15, batch_size=8)


def _series(steps=24 * 5 + 24 * 2 + 150, n=5, f=2, seed=3):
    rng = np.random.default_rng(seed)
    s = rng.standard_normal((steps, n, f)) * 7.0 + 20.0
    s[..., 1] = ((np.arange(steps) % 24) / 24.0)[:, None]
    return s


def test_core_matches_brute_force_windows():
    from multistgraph_amd.dataset import ResidentSeries
    raw = _series()
    rs = ResidentSeries(raw, **KW)
    rel = W.window_offsets(24, 2, 1, 1, 2, 5)
    assert np.array_equal(rs.rel, rel)
    starts = W.valid_label_starts(raw.shape[0], rel, 24)
    assert np.array_equal(rs.label_starts, starts)
    x, y = W.gather_windows(raw, starts, rel, 12)                      # what _generate_input_data materialises
    tr, ev, te = W.split_samples(len(starts), 0.7, 0.15)
    # StandardScaler(mean = x_train[..., :1].mean(), std = x_train[..., :1].std()) (traffic_state_datatset.py:911-913)
    assert abs(rs.scaler.mean - x[tr][..., :1].mean()) <= 1e-12 * abs(x[tr][..., :1].mean())
    assert abs(rs.scaler.std - x[tr][..., :1].std()) <= 1e-12 * x[tr][..., :1].std()
    # the resident float32 series holds what the reference's scaled float32 windows hold
    xs = x.copy()
    xs[..., :1] = rs.scaler.transform(xs[..., :1])
    host = rs.series_host.numpy()
    for part, idx in (("train", tr), ("eval", ev), ("test", te)):
        table = rs.part_table(part)
        padded = W.pad_with_last_sample(idx, 8)
        assert np.array_equal(table, starts[padded].astype(np.int32)) and len(table) % 8 == 0
        got = host[table[:, None] + rel[None, :]]
        assert np.array_equal(got, xs[padded].astype(np.float32))
    feat = rs.data_feature()
    assert (feat["len_closeness"], feat["len_period"], feat["len_trend"]) == (48, 24, 24)
    assert feat["feature_dim"] == 2 and feat["ext_dim"] == 1 and feat["num_batches"] == len(rs.loaders()[0])
    # test loader in order, train loader a permutation of its padded table
    te_batches = [b._host_starts for b in rs.loaders()[2]]
    assert np.array_equal(np.concatenate(te_batches), rs.part_table("test"))
    torch.manual_seed(1)
    tr_batches = np.concatenate([b._host_starts for b in rs.loaders()[0]])
    assert sorted(tr_batches.tolist()) == sorted(rs.part_table("train").tolist())
    assert not np.array_equal(tr_batches, rs.part_table("train"))


@pytest.mark.parametrize("kind", ["normal", "minmax01", "minmax11", "none", "log"])
def test_other_scalers_use_the_extrema_of_train_windows_and_labels(kind):
    from multistgraph_amd.dataset import ResidentSeries
    raw = np.abs(_series()) + 1.0
    rs = ResidentSeries(raw, scaler_type=kind, **KW)
    rel = W.window_offsets(24, 2, 1, 1, 2, 5)
    starts = W.valid_label_starts(raw.shape[0], rel, 24)
    x, y = W.gather_windows(raw, starts, rel, 12)
    tr, _, _ = W.split_samples(len(starts), 0.7, 0.15)
    if kind in ("normal", "minmax01", "minmax11"):
        assert rs.scaler.max == max(x[tr][..., :1].max(), y[tr][..., :1].max())
    if kind in ("minmax01", "minmax11"):
        assert rs.scaler.min == min(x[tr][..., :1].min(), y[tr][..., :1].min())
    xs = x.copy()
    xs[..., :1] = rs.scaler.transform(xs[..., :1])
    got = rs.series_host.numpy()[starts[:, None] + rel[None, :]]
    assert np.array_equal(got, xs.astype(np.float32))
    with pytest.raises(ValueError):
        ResidentSeries(raw, scaler_type="zscore", **KW)


def test_affine_scalers_are_detected_by_probing():
    """every LibCity scaler but LogScaler de-scales affinely: the fused loss / metric kernels take (mean, std)"""
    from multistgraph_amd import dataset as D
    from multistgraph_amd.model import MultiATGCN
    c = Case("tiny_multi_uni_c2")
    for sc, want in ((D.StandardScaler(3.0, 2.0), (3.0, 2.0)), (D.NoneScaler(), (0.0, 1.0)), (D.NormalScaler(5.0), (0.0, 5.0)),
                     (D.MinMax01Scaler(9.0, 1.0), (1.0, 8.0)), (D.MinMax11Scaler(9.0, 1.0), (5.0, 4.0)), (D.LogScaler(), None)):
        m = MultiATGCN(c.config(), dict(c.data_feature, scaler=sc))
        got = m._affine_scaler()
        assert (got is None) == (want is None)
        if want is not None:
            assert abs(got[0] - want[0]) < 1e-12 and abs(got[1] - want[1]) < 1e-12


def test_resident_batch_is_batch_compatible_and_refuses_the_host():
    from multistgraph_amd.dataset import ResidentSeries
    rs = ResidentSeries(_series(), **KW)
    b = next(iter(rs.loaders()[2]))
    assert len(b) == 8 and b.feature_name == {"X": "float", "y": "float"}
    assert "X" in b and "y" in b and "label_start" in b and "foo" not in b
    with pytest.raises(KeyError):
        b["y"]                                   # nothing is gathered before to_tensor
    with pytest.raises(RuntimeError):
        b.to_tensor(torch.device("cpu"))         # HIP path only: the series never lives on the host side of a batch
    with pytest.raises(TypeError):
        b.to_ndarray()
    with pytest.raises(KeyError):
        b["foo"]
    with pytest.raises(KeyError):
        b["foo"] = 1
    with pytest.raises(ValueError):              # the reference's sample filter guarantees input_window target rows only
        ResidentSeries(_series(), **dict(KW, output_window=30))
    with pytest.raises(ValueError):
        ResidentSeries(_series(steps=100), **KW)  # too short for the trend head (mth_dataset.py:134-137)


# ---- against the reference's own MTHDataset on synthetic atomic files (build container only) --------------------------
def _write_atomic_files(root, name, steps, n, seed):
    """a LibCity dataset directory: .geo / .rel / .dyna / .ext / .gbst in the format data_prepare/1.3*.py writes"""
    import pandas as pd
    rng = np.random.default_rng(seed)
    d = os.path.join(root, "raw_data", name)
    os.makedirs(d)
    geo_ids = np.arange(1000, 1000 + n)
    lon, lat = rng.uniform(-77.1, -76.9, n), rng.uniform(38.8, 39.0, n)
    pd.DataFrame({"geo_id": geo_ids, "type": "Point", "coordinates": ["[%.6f, %.6f]" % (a, b) for a, b in zip(lon, lat)]}
                 ).to_csv(os.path.join(d, name + ".geo"), index=False)
    w = rng.random((n, n))
    w[rng.random((n, n)) > 0.4] = 0.0
    w[np.diag_indices(n)] = 1.0 + w.sum(1)
    rows = [(i * n + j, "geo", geo_ids[i], geo_ids[j], w[i, j]) for i in range(n) for j in range(n)]
    pd.DataFrame(rows, columns=["rel_id", "type", "origin_id", "destination_id", "link_weight"]).to_csv(
        os.path.join(d, name + ".rel"), index=False)
    times = pd.date_range("2019-01-01", periods=steps, freq="h").strftime("%Y-%m-%dT%H:%M:%SZ")
    visits = rng.standard_normal((n, steps)) * 3.0 + 10.0
    pd.DataFrame({"dyna_id": np.arange(n * steps), "type": "state", "time": np.tile(times, n),
                  "entity_id": np.repeat(geo_ids, steps), "Visits": visits.reshape(-1)}).to_csv(
        os.path.join(d, name + ".dyna"), index=False)
    pd.DataFrame({"ext_id": np.arange(steps), "time": times, "holiday": rng.integers(0, 2, steps),
                  "weekend": rng.integers(0, 2, steps), "temp": rng.standard_normal(steps)}).to_csv(
        os.path.join(d, name + ".ext"), index=False)
    pd.DataFrame({"geo_id": geo_ids, "All_m": rng.uniform(5, 50, n), "All_std": rng.uniform(2, 30, n)}).to_csv(
        os.path.join(d, name + ".gbst"), index=False)


class _Config(dict):
    """ConfigParser-shaped: get / [] / []= / in (libcity/config/config_parser.py:134-151)"""


@pytest.mark.parametrize("load_dynamic", [False, True])
def test_resident_dataset_equals_the_reference_dataset(tmp_path, monkeypatch, load_dynamic):
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "libcity")):
        pytest.skip("reference checkout not present (GPU box): the dataset comparison runs in the build container")
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)
    if not hasattr(np, "float"):      # the reference was written against numpy < 1.24 (traffic_state_datatset.py:284)
        monkeypatch.setattr(np, "float", float, raising=False)
    from libcity.data.dataset.dataset_subclass.mth_dataset import MTHDataset
    import multistgraph_amd.dataset as D
    D = importlib.reload(D)           # pick up the reference base class if the module was imported without it
    assert issubclass(D.MTHDatasetResident, MTHDataset)
    name = "SYNTH_SG"
    _write_atomic_files(str(tmp_path), name, steps=24 * 5 + 24 * 2 + 130, n=6, seed=11)
    monkeypatch.chdir(tmp_path)       # the reference reads ./raw_data/<dataset>/ and writes ./libcity/cache/
    # pad_with_last_sample = False here: the reference pads with np.repeat over a list of ragged (x, y) tuples
    # (data/utils.py:53-61), which numpy >= 1.24 refuses; the padding rule itself is pinned to the reference's function on
    # index arrays by tests/golden/windows_small.npz (tests/test_windows.py)
    cfg = dict(dataset=name, batch_size=8, cache_dataset=False, num_workers=0, pad_with_last_sample=False, train_rate=0.7,
               eval_rate=0.15, scaler="standard", ext_scaler="none", load_external=True, load_dynamic=load_dynamic,
               normal_external=False, add_time_in_day=True, add_day_in_week=False, input_window=24, output_window=12,
               use_3tu=True, groupstd=True, add_static=False, len_closeness=2, len_period=1, len_trend=1,
               interval_period=2, interval_trend=5, hour_each_day=24, data_col=["Visits"], weight_col="link_weight",
               ext_col=["holiday", "weekend", "temp"], data_files=[name], geo_file=name, rel_file=name, ext_file=name,
               output_dim=1, time_intervals=3600, init_weight_inf_or_zero="zero", set_weight_link_or_dist="dist",
               calculate_weight_adj=False)
    ref = MTHDataset(_Config(cfg))
    ref_loaders = ref.get_data()
    # the registry change of INTEGRATION.md (one import line in dataset_subclass/__init__.py), then the reference's own
    # resolver: get_dataset looks config['dataset_class'] up by name (libcity/data/utils.py:10-28)
    import libcity.data.dataset.dataset_subclass as sub
    from libcity.data.utils import get_dataset
    monkeypatch.setattr(sub, "MTHDatasetResident", D.MTHDatasetResident, raising=False)
    mine = get_dataset(_Config(dict(cfg, dataset_class="MTHDatasetResident")))
    assert type(mine) is D.MTHDatasetResident
    my_loaders = mine.get_data()
    rf, mf = ref.get_data_feature(), mine.get_data_feature()
    assert set(rf) == set(mf)
    for k in ("num_nodes", "feature_dim", "output_dim", "ext_dim", "len_closeness", "len_period", "len_trend",
              "num_batches"):
        assert rf[k] == mf[k], k
    assert np.array_equal(rf["adj_mx"], mf["adj_mx"]) and rf["static"] is None and mf["static"] is None
    assert rf["ct_visit_mstd"].equals(mf["ct_visit_mstd"]) and rf["coordinate"].equals(mf["coordinate"])
    assert abs(rf["scaler"].mean - mf["scaler"].mean) <= 1e-12 * abs(rf["scaler"].mean)
    assert abs(rf["scaler"].std - mf["scaler"].std) <= 1e-12 * rf["scaler"].std
    assert mf["feature_dim"] == (5 if load_dynamic else 2)
    host, rel, out = mine.core.series_host.numpy(), mine.core.rel, 12
    for which, seed in ((2, None), (0, 123), (1, 7)):       # test (in order), train and eval (shuffled by torch's RNG)
        assert len(ref_loaders[which]) == len(my_loaders[which])
        if seed is not None:
            torch.manual_seed(seed)
        ref_batches = []
        for b in ref_loaders[which]:
            b.to_tensor(torch.device("cpu"))
            ref_batches.append((b["X"].numpy(), b["y"].numpy()))
        if seed is not None:
            torch.manual_seed(seed)
        for (rx, ry), b in zip(ref_batches, my_loaders[which]):
            ls = b._host_starts.astype(np.int64)
            assert np.array_equal(host[ls[:, None] + rel[None, :]], rx)                 # every window, bit for bit
            assert np.array_equal(host[ls[:, None] + np.arange(out)[None, :]], ry)      # every label block


# ---- GPU: one epoch over the resident loaders = the window path ---------------------------------------------------------
@pytest.mark.gpu
def test_one_epoch_over_the_resident_dataset(lib_built, monkeypatch):
    """The executor's loops driven unchanged (traffic_state_executor.py:398-448, 252-290): for batch in loader:
    batch.to_tensor(device); model.calculate_loss(batch) / model.predict(batch); batch['y'].  Losses, predictions and the
    evaluator table equal those of the SAME samples fed as materialised windows, and the window path equals the oracle."""
    from multistgraph_amd.dataset import ResidentSeries
    from multistgraph_amd.evaluator import ALLOWED_METRICS, DeviceEvaluator
    from multistgraph_amd.model import MultiATGCN
    from oracle import matgcn_oracle as O
    c = Case("tiny_multi_uni_out12")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4)
    raw = rng.standard_normal((24 * 28 + 24 * 2 + 60, c.n, c.feat)) * 5.0 + 12.0
    raw[..., 1] = ((np.arange(raw.shape[0]) % 24) / 24.0)[:, None]
    raw[rng.random(raw.shape[:2]) < 0.02, 0] = 0.0
    rs = ResidentSeries(raw, input_window=24, output_window=c.out, batch_size=c.b, train_rate=0.7, eval_rate=0.15)
    m = MultiATGCN(c.config("cuda:0"), dict(c.data_feature, scaler=rs.scaler)).to(dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})
    train, evl, test = rs.loaders()
    host = rs.series_host.numpy()
    p = O.to_tensors(c.state)
    st = O.supports_as_tensors(c.gold["static_supports"])
    # validation epoch (no_grad, eval): loss per batch, series path vs window path vs oracle
    m.eval()
    torch.manual_seed(0)
    with torch.no_grad():
        for i, batch in enumerate(evl):
            batch.to_tensor(dev)
            loss = m.calculate_loss(batch)
            ls = batch._host_starts.astype(np.int64)
            x = host[ls[:, None] + rs.rel[None, :]]
            y = host[ls[:, None] + np.arange(c.out)[None, :]]
            win = {"X": torch.from_numpy(x).to(dev), "y": torch.from_numpy(y).to(dev)}
            assert torch.equal(batch["y"], win["y"]) and torch.equal(batch["X"], win["X"])
            assert torch.equal(m.predict(batch), m.predict(win))
            assert abs(float(loss) - float(m.calculate_loss(win))) <= 1e-6 * abs(float(loss))
            if i == 0:
                want = O.calculate_loss(torch.from_numpy(x), torch.from_numpy(y), p, st, c.oracle_cfg(),
                                        rs.scaler.mean, rs.scaler.std, faithful=False)
                assert abs(float(loss) - float(want)) <= 1e-4 * abs(float(want))
            if i >= 2:
                break
    # test epoch through the device evaluator: no prediction leaves the GPU
    ev = DeviceEvaluator({"metrics": list(ALLOWED_METRICS), "evaluator_mode": "single"}, streaming=True)
    preds, trues = [], []
    with torch.no_grad():
        for i, batch in enumerate(test):
            batch.to_tensor(dev)
            pred = m.collect_metrics(ev, batch)
            preds.append(rs.scaler.inverse_transform(pred.cpu()))
            trues.append(rs.scaler.inverse_transform(batch["y"][..., 0:1].cpu()))
            if i >= 3:
                break
    res = ev.evaluate()
    want = O.evaluator_table(torch.cat(preds), torch.cat(trues), "single")
    for k, v in want.items():
        assert (np.isinf(v) and np.isinf(res[k])) or abs(res[k] - v) <= 1e-4 * abs(v) + 1e-7, k
    # a training step: the series path and the same samples as materialised windows give the same loss and gradients;
    # then a few optimizer steps on that batch: the loss moves down (gradients flow through the series path)
    m.train()
    monkeypatch.setattr(torch.nn.functional, "dropout", lambda inp, p=0.5, training=True, inplace=False: inp)
    opt = torch.optim.Adam(m.parameters(), lr=2e-3)
    torch.manual_seed(0)
    batch = next(iter(train))
    batch.to_tensor(dev)
    win = {"X": batch["X"].clone(), "y": batch["y"].clone()}
    opt.zero_grad()
    loss = m.calculate_loss(batch)
    loss.backward()
    g_series = {k: q.grad.clone() for k, q in m.named_parameters() if q.grad is not None}
    opt.zero_grad()
    lw = m.calculate_loss(win)
    lw.backward()
    assert abs(float(lw.detach()) - float(loss.detach())) <= 1e-6 * abs(float(loss.detach()))
    for k, q in m.named_parameters():
        if q.grad is not None:
            assert max_norm_err(q.grad.cpu().numpy(), g_series[k].cpu().numpy()) <= 1e-5, k
    losses = []
    for _ in range(6):
        opt.zero_grad()
        step_loss = m.calculate_loss(batch)
        step_loss.backward()
        opt.step()
        losses.append(float(step_loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
