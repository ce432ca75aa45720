"""Wall time per training step of a FREE-RUNNING loop (no synchronisation between steps - what an executor's epoch does)
next to the device time of synchronised steps.  usage: train_loop_wall.py [workload] [batch] [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from multistgraph_amd import synthetic as syn
name = sys.argv[1] if len(sys.argv) > 1 else "bm403"
w = dict(bench.WORKLOADS[name])
if len(sys.argv) > 2 and int(sys.argv[2]) > 0:
    w["batch"] = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")
if os.environ.get("MATGCN_PG") == "1":        # inside a one-rank RCCL process group, like a data-parallel job
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dist.barrier()
if os.environ.get("MATGCN_POOL") == "1":      # the data-parallel jobs' stream mode (matgcn_set_stream_pool) in a single process
    from multistgraph_amd import sharding
    sharding.use_own_stream_pool()
model, df, cfg = bench.build_model(w, dev, 0)
model.train()
x_np, y_np = syn.make_batch_arrays(w["batch"], w["nodes"], w["out"], 0, feat=2)
batch = {"X": torch.from_numpy(x_np).to(dev), "y": torch.from_numpy(y_np).to(dev)}
opt = torch.optim.Adam(model.parameters(), lr=1e-3)


def step():
    opt.zero_grad()
    loss = model.calculate_loss(batch)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
# synchronised steps: device time per step
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dts = []
for _ in range(10):
    e0.record(); step(); e1.record(); torch.cuda.synchronize()
    dts.append(e0.elapsed_time(e1))
dts.sort()
# free-running
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("%s B=%d queues=%s pool=%s pg=%s  synchronised step (events) median %.2f ms | free-running: wall %.2f ms per step, host enqueue %.2f ms per step" % (
    name, w["batch"], os.environ.get("GPU_MAX_HW_QUEUES"), os.environ.get("MATGCN_POOL", "0"), os.environ.get("MATGCN_PG", "0"), dts[len(dts) // 2], t_all / steps * 1e3, t_host / steps * 1e3), flush=True)
