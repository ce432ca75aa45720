"""Two serial-schedule forwards of the bench workload (Baltimore 403, B=64) - the target of the PMC passes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from multistgraph_amd import _lib, synthetic as syn
w = dict(bench.WORKLOADS["bm403"])
dev = torch.device("cuda:0")
model, df, cfg = bench.build_model(w, dev, 0)
_lib.load().matgcn_set_wavefront(0)
x_np, _ = syn.make_batch_arrays(w["batch"], w["nodes"], w["out"], 0, feat=2)
batch = {"X": torch.from_numpy(x_np).to(dev)}
with torch.no_grad():
    for i in range(2):
        model.predict(batch)
        torch.cuda.synchronize()
        print("forward", i, "done", flush=True)
