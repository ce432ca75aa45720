"""A short training run on the HIP path, driven the way TrafficStateExecutor drives the reference
(model.train(); loss = model.calculate_loss(batch); loss.backward(); Adam; model.eval() validation under no_grad):
a synthetic hourly series with daily + weekly structure on the DC-sized graph (237 nodes), windows in the reference's
multi-temporal-head layout (multistgraph_amd/windows.py).  Prints the loss curve and MAE@k before / after.
usage: train_demo.py [steps]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from multistgraph_amd import windows

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
w = dict(bench.WORKLOADS["dc237"])
dev = torch.device("cuda:0")
torch.manual_seed(0)
model, df, cfg = bench.build_model(w, dev, 0)
n, out, B = w["nodes"], w["out"], w["batch"]
rng = np.random.default_rng(0)
hours = 24 * 7 * 10
t = np.arange(hours)
amp, phase = rng.uniform(0.5, 1.5, n), rng.uniform(0, 2 * np.pi, n)
flow = (amp[None, :] * np.sin(2 * np.pi * t[:, None] / 24 + phase[None, :]) +
        0.5 * np.sin(2 * np.pi * t[:, None] / (24 * 7) + 0.3 * phase[None, :]) + 0.2 * rng.standard_normal((hours, n)))
series = np.stack([flow, np.tile(((t % 24) / 24.0)[:, None], (1, n))], -1).astype(np.float32)
rel = windows.window_offsets(24)
starts = windows.valid_label_starts(hours, rel, 24)
split = int(0.8 * len(starts))


def batch_of(idx):
    x, y = windows.gather_windows(series, starts[idx], rel, out)
    return {"X": torch.from_numpy(x).to(dev), "y": torch.from_numpy(y).to(dev)}


val = batch_of(np.arange(split, split + B))


def evaluate():
    model.eval()
    with torch.no_grad():
        return [float(v) for v in model.horizon_mae(val)]


before = evaluate()
opt = torch.optim.Adam(model.parameters(), lr=3e-3)
losses = []
for i in range(steps):
    model.train()
    batch = batch_of(rng.choice(split, B, replace=False))
    opt.zero_grad()
    loss = model.calculate_loss(batch)
    loss.backward()
    opt.step()
    losses.append(float(loss.detach()))
    if i % 10 == 0 or i == steps - 1:
        print("step %3d  train loss %.4f" % (i, losses[-1]), flush=True)
after = evaluate()
print("validation MAE@1/6/12 before: %.4f %.4f %.4f" % (before[0], before[5], before[-1]))
print("validation MAE@1/6/12 after : %.4f %.4f %.4f" % (after[0], after[5], after[-1]))
assert after[0] < before[0] and losses[-1] < losses[0]
