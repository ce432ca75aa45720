#!/usr/bin/env python3
"""Where does the HOST spend its time in a B = 16 training step?  cProfile over free-running steps (DC 237)."""
import cProfile, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from multistgraph_amd import synthetic as syn
name = sys.argv[1] if len(sys.argv) > 1 else "dc237"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
w = dict(bench.WORKLOADS[name], batch=batch)
dev = torch.device("cuda:0")
model, _, _ = bench.build_model(w, dev, 0)
x_np, y_np = syn.make_batch_arrays(batch, w["nodes"], w["out"], 0, feat=2)
b = {"X": torch.from_numpy(x_np).to(dev), "y": torch.from_numpy(y_np).to(dev)}
model.train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
def step():
    opt.zero_grad()
    model.calculate_loss(b).backward()
    opt.step()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(40):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
