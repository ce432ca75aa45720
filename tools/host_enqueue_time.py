"""Is the training step bound by the HOST's launch rate?  Times, per step and WITHOUT a profiler attached, how long the
host needs to enqueue the forward and the backward (perf_counter around the calls, no synchronisation inside) next to
the device time of the same step (HIP events).  usage: host_enqueue_time.py [workload] [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from multistgraph_amd import synthetic as syn
name = sys.argv[1] if len(sys.argv) > 1 else "bm403"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w = dict(bench.WORKLOADS[name])
dev = torch.device("cuda:0")
model, df, cfg = bench.build_model(w, dev, 0)
model.train()
x_np, y_np = syn.make_batch_arrays(w["batch"], w["nodes"], w["out"], 0, feat=2)
batch = {"X": torch.from_numpy(x_np).to(dev), "y": torch.from_numpy(y_np).to(dev)}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for i in range(steps):
    model.zero_grad()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    loss = model.calculate_loss(batch)
    ev[1].record()
    t1 = time.perf_counter()
    loss.backward()
    ev[2].record()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("step %d  host enqueue: forward %.2f ms, backward %.2f ms (returned %.2f ms after the step began); device: forward "
          "%.2f ms, backward %.2f ms; step done at %.2f ms" % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3,
                                                               ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]),
                                                               (t3 - t0) * 1e3), flush=True)
