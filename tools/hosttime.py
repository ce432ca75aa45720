import sys, time, torch
sys.path.insert(0, '.')
import bench
w = dict(bench.WORKLOADS["bm403"])
dev = torch.device("cuda:0")
model, df, cfg = bench.build_model(w, dev, 0)
from multistgraph_amd import synthetic as syn
x_np, y_np = syn.make_batch_arrays(w["batch"], w["nodes"], w["out"], 0, feat=2)
batch = {"X": torch.from_numpy(x_np).to(dev)}
with torch.no_grad():
    for _ in range(3): model.predict(batch)
    torch.cuda.synchronize()
    for cache in (False, True):
        model.cache_prepared = cache
        model.predict(batch); torch.cuda.synchronize()
        host = []
        t0 = time.perf_counter()
        for _ in range(10):
            a = time.perf_counter(); model.predict(batch); host.append(time.perf_counter() - a)
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        print("cache_prepared", cache, "wall/forward ms %.3f" % (tot / 10 * 1e3), "host enqueue ms/forward %.3f" % (sum(host) / 10 * 1e3))
