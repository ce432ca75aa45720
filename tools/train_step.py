"""Training steps of the bench workload (default Baltimore 403, B=64): forward_train + backward through the plugin
surface (calculate_loss().backward()), timed with HIP events.  usage: train_step.py [workload] [steps] [serial|wave] [batch]
(serial: matgcn_set_wavefront(0) - every kernel alone on one stream, so a profiler's durations are the kernels' own)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from multistgraph_amd import synthetic as syn
name = sys.argv[1] if len(sys.argv) > 1 else "bm403"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = dict(bench.WORKLOADS[name])
if len(sys.argv) > 4:
    w["batch"] = int(sys.argv[4])      # e.g. 16, the reference's shipped batch_size
dev = torch.device("cuda:0")
model, df, cfg = bench.build_model(w, dev, 0)
model.train()
if len(sys.argv) > 3 and sys.argv[3] == "serial":
    from multistgraph_amd import _lib
    _lib.load().matgcn_set_wavefront(0)
x_np, y_np = syn.make_batch_arrays(w["batch"], w["nodes"], w["out"], 0, feat=2)
batch = {"X": torch.from_numpy(x_np).to(dev), "y": torch.from_numpy(y_np).to(dev)}
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for i in range(steps):
    opt.zero_grad()
    ev[0].record()
    loss = model.calculate_loss(batch)
    ev[1].record()
    loss.backward()
    ev[2].record()
    opt.step()
    ev[3].record()
    torch.cuda.synchronize()
    print("step %d loss %.5f  forward %.2f ms  backward %.2f ms  adam %.2f ms" % (
        i, float(loss), ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])), flush=True)
print("train buffer %.2f GB  workspace %.2f GB  peak torch memory %.2f GB" % (
    next(iter(model._paths.values()))._train.numel() * 4 / 2**30,
    next(iter(model._paths.values())).workspace.numel() * 4 / 2**30, torch.cuda.max_memory_allocated() / 2**30))
