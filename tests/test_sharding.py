"""N > 1 path on CPU: world_size-2 gloo processes exercise the batch sharding, the max-over-ranks throughput
rule of bench.py, the prediction gather and the flat gradient bucket."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multistgraph_amd import sharding as sh


def test_shard_bounds_cover_the_batch():
    for gb in (0, 1, 5, 64, 70, 512):
        for world in (1, 2, 3, 8):
            spans = [sh.shard_bounds(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sh.shard_bounds(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gb = 7                                   # ragged: 4 + 3
        rng = np.random.default_rng(3)
        x = torch.from_numpy(rng.standard_normal((gb, 5, 4, 2)).astype(np.float32))
        y = torch.from_numpy(rng.standard_normal((gb, 3, 4, 2)).astype(np.float32))
        mine = sh.shard_batch({"X": x, "y": y}, rank, world)
        lo, hi = sh.shard_bounds(gb, rank, world)
        assert mine["X"].shape[0] == hi - lo and torch.equal(mine["y"], y[lo:hi])
        # a stand-in "forward": any per-sample function shards exactly
        pred_local = mine["X"].sum(dim=(1, 3), keepdim=False)
        full = sh.gather_predictions(pred_local, gb)
        assert torch.equal(full, x.sum(dim=(1, 3)))
        # throughput rule: sum of units over the slowest rank's time
        rate, tmax = sh.job_throughput(100.0 * (rank + 1), 0.5 * (rank + 1))
        assert abs(tmax - 0.5 * world) < 1e-12 and abs(rate - 100.0 * world * (world + 1) / 2 / (0.5 * world)) < 1e-9
        # flat gradient bucket = mean over ranks, shapes preserved
        g1 = torch.full((3, 2), float(rank + 1))
        g2 = torch.arange(4, dtype=torch.float32) * (rank + 1)
        sh.flat_allreduce_mean_([g1, None, g2])
        mean = (world + 1) / 2
        assert torch.allclose(g1, torch.full((3, 2), mean)) and torch.allclose(g2, torch.arange(4.0) * mean)
        # the gradients as VIEWS of one flat buffer (what matgcn_backward fills): one collective on the buffer, and
        # every parameter's .grad - adopted by autograd from those views - sees the mean without a copy back
        bucket = torch.zeros(192)

        class _Step(torch.autograd.Function):
            @staticmethod
            def forward(ctx, a, b):
                return (a.sum() + b.sum()) * 0.0 + 1.0

            @staticmethod
            def backward(ctx, up):
                va, vb = bucket[0:6].view(2, 3), bucket[64:68].view(4)
                va.fill_(float(rank + 1)); vb.copy_(torch.arange(4.0) * (rank + 1))
                return va, vb

        pa, pb = torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(4))
        _Step.apply(pa, pb).backward()
        lo, hi = bucket.data_ptr(), bucket.data_ptr() + bucket.numel() * 4
        assert lo <= pa.grad.data_ptr() < hi and lo <= pb.grad.data_ptr() < hi      # adopted, not copied
        sh.bucket_allreduce_mean_(bucket)
        assert torch.allclose(pa.grad, torch.full((2, 3), mean)) and torch.allclose(pb.grad, torch.arange(4.0) * mean)
        assert float(bucket[6:64].abs().max()) == 0.0                               # alignment gaps stay zero
        # replica check
        assert sh.replicas_in_sync([torch.ones(3)])
        assert not sh.replicas_in_sync([torch.ones(3) * (rank + 1)])
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))
