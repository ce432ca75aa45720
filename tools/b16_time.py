#!/usr/bin/env python3
"""The shipped batch size (16) on the bench graphs: bench.small_batch_times as a standalone timer.
    python tools/b16_time.py [workload ...] [--batch 16]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
args = [a for a in sys.argv[1:] if not a.startswith("--")]
batch = 16
for a in sys.argv[1:]:
    if a.startswith("--batch="):
        batch = int(a.split("=")[1])
for name in args or ["bm403", "dc237"]:
    r = bench.small_batch_times(bench.WORKLOADS[name], torch.device("cuda:0"), batch=batch)
    r.pop("note")
    print(name, json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}, default=lambda o: round(o, 3)), flush=True)
