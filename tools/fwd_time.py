#!/usr/bin/env python3
"""Lab helper: time the forward (and optionally the training step) of a bench workload in THIS process' library
configuration (env knobs such as MATGCN_CU_SPLIT / MATGCN_LIB are read by the library at first use).
    python tools/fwd_time.py [--workload bm403] [--iters 60] [--train] [--tag text]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="bm403")
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--serial", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (16 = the reference's shipped batch_size)")
    ap.add_argument("--token", action="store_true", help="matgcn_set_wavefront(2): the graph mixes of all chains in one global order")
    ap.add_argument("--cache-prepared", action="store_true")
    ap.add_argument("--split", type=int, default=0, help="matgcn_set_batch_split(n)")
    ap.add_argument("--kernels", action="store_true", help="per-kernel launch averages, wavefront off (HIP events)")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    from multistgraph_amd import _lib, synthetic as syn
    dev = torch.device("cuda", 0)
    w = dict(bench.WORKLOADS[args.workload])
    if args.batch:
        w["batch"] = args.batch
    model, df, cfg = bench.build_model(w, dev, 0)
    model.cache_prepared = bool(args.cache_prepared)
    if os.environ.get("MATGCN_POOL") == "1":      # the data-parallel jobs' stream mode (matgcn_set_stream_pool) in a single process
        _lib.check(_lib.load().matgcn_set_stream_pool(1), "matgcn_set_stream_pool")
    if args.serial:
        _lib.load().matgcn_set_wavefront(0)
    if args.token:
        _lib.load().matgcn_set_wavefront(2)
    if args.split:
        _lib.load().matgcn_set_batch_split(args.split)
    x_np, y_np = syn.make_batch_arrays(w["batch"], w["nodes"], w["out"], 0, feat=2)
    batch = {"X": torch.from_numpy(x_np).to(dev), "y": torch.from_numpy(y_np).to(dev)}
    with torch.no_grad():
        for _ in range(10):
            model.predict(batch)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
        for e0, e1 in evs:
            e0.record(); model.predict(batch); e1.record()
        torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    line = "%-28s %s fwd median %.3f ms  p10 %.3f  p90 %.3f" % (args.tag, args.workload, statistics.median(ts),
                                                              ts[len(ts) // 10], ts[len(ts) * 9 // 10])
    if args.kernels:
        ser = bench.in_situ_kernel_times(model, batch, wavefront=False, forwards=2)
        line += "   serial us/launch: " + "  ".join("%s %.1f (x%d)" % (k, 1e3 * statistics.mean(v), len(v) // 2)
                                                    for k, v in sorted(ser.items()))
        line += "   serial sum %.2f ms" % (sum(sum(v) for v in ser.values()) / 2)
    if args.train:
        t = bench.train_step_times(model, batch, w, warm=3, steps=int(os.environ.get("TRAIN_STEPS", "8")))
        line += "   train: fwd %.2f bwd %.2f step %.2f ms" % (t["forward_ms"], t["backward_ms"], t["ms_per_step"])
    print(line, flush=True)


if __name__ == "__main__":
    main()
