"""matgcn_prepare alone, timed with events (bm403, B=64)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
w = dict(bench.WORKLOADS["bm403"]); dev = torch.device("cuda:0")
model, df, cfg = bench.build_model(w, dev, 0)
from multistgraph_amd import synthetic as syn
x_np, _ = syn.make_batch_arrays(w["batch"], w["nodes"], w["out"], 0, feat=2)
with torch.no_grad():
    model.predict({"X": torch.from_numpy(x_np).to(dev)})
hp = next(iter(model._paths.values()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    for _ in range(20): hp.prepare()
    e1.record(); torch.cuda.synchronize()
    print("prepare %.3f ms" % (e0.elapsed_time(e1) / 20))
