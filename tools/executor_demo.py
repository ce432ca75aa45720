"""The reference's executor loops over the device-resident dataset, and what the
resident loader saves per training step.

Drives the plugin class exactly the way TrafficStateExecutor does (traffic_state_executor.py:325-448, 252-323): epochs of
`for batch in train_loader: zero_grad; batch.to_tensor(device); loss = model.calculate_loss(batch); loss.backward();
clip_grad_norm_(5); step`, a validation epoch under no_grad, Adam + MultiStepLR, then the test set through the
evaluator - here ResidentSeries loaders and the DeviceEvaluator, so no window, label or prediction crosses PCIe.
Then times the SAME training steps fed the reference's way: float64 windows materialised on the host
(mth_dataset.py:110-160), a deep copy per item in the collate function (data/utils.py:68-72), FloatTensor + H2D per batch
(batch.py:43-57).
usage: executor_demo.py [epochs] [steps_per_epoch]"""
import copy
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from multistgraph_amd import windows  # noqa: E402
from multistgraph_amd.dataset import ResidentSeries  # noqa: E402
from multistgraph_amd.evaluator import ALLOWED_METRICS, DeviceEvaluator, groupstd_table  # noqa: E402
from multistgraph_amd.model import MultiATGCN  # noqa: E402
from multistgraph_amd import synthetic as syn  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps_per_epoch = int(sys.argv[2]) if len(sys.argv) > 2 else 24
w = dict(bench.WORKLOADS["dc237"])
dev = torch.device("cuda:0")
n, out, B = w["nodes"], w["out"], w["batch"]
rng = np.random.default_rng(0)
hours = 3624                                        # the length of the reference's hourly series (README.md:48)
t = np.arange(hours)
amp, phase = rng.uniform(20.0, 60.0, n), rng.uniform(0, 2 * np.pi, n)
visits = np.maximum(0.0, 30.0 + amp[None, :] * np.sin(2 * np.pi * t[:, None] / 24 + phase[None, :]) +
                    10.0 * np.sin(2 * np.pi * t[:, None] / (24 * 7)) + 5.0 * rng.standard_normal((hours, n)))
raw = np.stack([visits, np.tile(((t % 24) / 24.0)[:, None], (1, n))], -1)      # float64, as _load_dyna returns it

rs = ResidentSeries(raw, input_window=24, output_window=out, batch_size=B, train_rate=0.7, eval_rate=0.15)
train, evl, test = rs.loaders()
df = dict(syn.make_data_feature(n, 0, w["city"]), scaler=rs.scaler)
cfg = dict(input_window=24, output_window=out, add_time_in_day=True, add_day_in_week=False, load_dynamic=False,
           adjtype="multi", adpadj="unidirection", cheb_order=2, embed_dim_node=20, embed_dim_adj=20, rnn_units=64,
           num_layers=2, device=dev, batch_size=B)
torch.manual_seed(0)
model = MultiATGCN(cfg, df).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.003, eps=1e-8)            # TrafficStateExecutor.json defaults
sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[5, 10, 20, 30], gamma=0.75)
print("series %s float64 -> resident float32 %.1f MB; %d / %d / %d batches of %d; scaler mean %.3f std %.3f" % (
    raw.shape, rs.series_host.numel() * 4 / 1e6, len(train), len(evl), len(test), B, rs.scaler.mean, rs.scaler.std))

for epoch in range(epochs):
    model.train()
    losses = []
    t0 = time.perf_counter()
    for i, batch in enumerate(train):
        if i >= steps_per_epoch:
            break
        opt.zero_grad()
        batch.to_tensor(dev)
        loss = model.calculate_loss(batch)
        losses.append(loss.item())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
        opt.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sched.step()
    with torch.no_grad():
        model.eval()
        vl = []
        for i, batch in enumerate(evl):
            if i >= 4:
                break
            batch.to_tensor(dev)
            vl.append(model.calculate_loss(batch).item())
    print("epoch %d  train loss %.4f -> %.4f  valid loss %.4f  %.1f ms per training step (resident loader)" % (
        epoch, losses[0], losses[-1], float(np.mean(vl)), dt / len(losses) * 1e3), flush=True)
resident_ms = dt / len(losses) * 1e3

# ---- test set through the device evaluator + the group-std re-transform table
ev = DeviceEvaluator({"metrics": list(ALLOWED_METRICS), "evaluator_mode": "single"}, streaming=True)
all_m, all_std = rng.uniform(5, 60, n).astype(np.float32), rng.uniform(3, 40, n).astype(np.float32)
sums = None
with torch.no_grad():
    model.eval()
    for i, batch in enumerate(test):
        if i >= 4:
            break
        batch.to_tensor(dev)
        pred = model.collect_metrics(ev, batch)
        _, sums = groupstd_table(pred, batch["series"], all_m, all_std, 0, rs.scaler.mean, rs.scaler.std,
                                 label_start=batch["label_start"], sums=sums)
res = ev.evaluate()
print("test MAE@1/6/12 %.3f %.3f %.3f   masked_MAPE@12 %.4f   R2@12 %.4f" % (
    res["MAE@1"], res["MAE@6"], res["MAE@%d" % out], res["masked_MAPE@%d" % out], res["R2@%d" % out]))

# ---- the same training steps fed the reference's way: host windows, deep-copy collate, H2D per batch
rel = rs.rel
starts = rs.label_starts[rs.parts["train"]][: steps_per_epoch * B]
scaled = np.array(raw, dtype=np.float64)
scaled[..., :1] = rs.scaler.transform(scaled[..., :1])
t0 = time.perf_counter()
xs, ys = windows.gather_windows(scaled, starts, rel, out)                # (samples, 96, N, F) float64, built once
build_s = time.perf_counter() - t0
items = list(zip(xs, ys))
model.train()
t0 = time.perf_counter()
host_s = 0.0
for i in range(steps_per_epoch):
    h0 = time.perf_counter()
    chunk = [copy.deepcopy(it) for it in items[i * B:(i + 1) * B]]       # the collate function's deep copy
    X = torch.FloatTensor(np.array([c[0] for c in chunk])).to(dev)       # Batch.to_tensor
    y = torch.FloatTensor(np.array([c[1] for c in chunk])).to(dev)
    host_s += time.perf_counter() - h0
    opt.zero_grad()
    loss = model.calculate_loss({"X": X, "y": y})
    loss.item()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
    opt.step()
torch.cuda.synchronize()
win_ms = (time.perf_counter() - t0) / steps_per_epoch * 1e3
print("window loader: %.1f ms per training step (of which %.1f ms host collate + H2D of %.1f MB per batch; windows of %d "
      "samples built once in %.2f s, %.2f GB float64) vs resident loader %.1f ms per step" % (
          win_ms, host_s / steps_per_epoch * 1e3, (X.numel() + y.numel()) * 4 / 1e6, len(items), build_s,
          (xs.nbytes + ys.nbytes) / 1e9, resident_ms))
