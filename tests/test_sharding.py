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


# ---- the one data-parallel entry point: sharding.allreduce_model_grads_(model) ------------------------------------------
def _fake_backward(model, rank, with_bucket):
    """What loss.backward() leaves behind on the HIP path, imitated on the CPU: the gradients of the HIP-path
    parameters as views of one flat buffer (HotPath.grad_bucket), the host-side static_initial_* layers' gradients as
    tensors of their own (they arrive through d_h0 and torch autograd); without a bucket every gradient is its own
    tensor (rnn_units < 64: autograd slices the padded gradients back).  Values depend on the rank."""
    import types
    named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
    hip = [(k, p) for k, p in named if not k.startswith("static_initial")]
    gen = torch.Generator().manual_seed(100 + rank)
    if with_bucket:
        offs, total = {}, 0
        for k, p in hip:
            offs[k] = total
            total += (p.numel() + 63) // 64 * 64
        bucket = torch.zeros(total)
        for k, p in hip:
            view = bucket[offs[k]:offs[k] + p.numel()].view(p.shape)
            view.copy_(torch.randn(p.shape, generator=gen))
            p.grad = view
        model._paths = {2: types.SimpleNamespace(grad_bucket=bucket)}
    else:
        for k, p in hip:
            p.grad = torch.randn(p.shape, generator=gen)
        model._paths = {}
    for k, p in named:
        if k.startswith("static_initial_gru"):      # static_initial_node is never used by forward: no gradient
            p.grad = torch.randn(p.shape, generator=gen)
    return {k: p.grad.clone() for k, p in named if p.grad is not None}


def _dp_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import Case
    from multistgraph_amd.model import MultiATGCN
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for name, with_bucket in (("tiny_multi_uni_c2_static", True), ("hid32_multi_uni_c2", False),
                                  ("hid32_multi_uni_c2_static", False), ("tiny_multi_uni_c2", True)):
            c = Case(name)
            torch.manual_seed(5)                                    # same initial weights on every rank
            model = MultiATGCN(c.config(), c.data_feature)
            mine = _fake_backward(model, rank, with_bucket)
            every = [_fake_backward(MultiATGCN(c.config(), c.data_feature), r, False) for r in range(world)]
            bucket, rest = model.gradient_exchange()
            assert (bucket is not None) == with_bucket
            if c.static_dim > 0:                                    # the pitfall of round 2: these live OUTSIDE the bucket
                assert rest and model.gradient_bucket() is None
            elif with_bucket:
                assert not rest and model.gradient_bucket() is bucket
            info = sh.allreduce_model_grads_(model)
            assert info["bucket"] == with_bucket
            for k, p in model.named_parameters():
                if p.grad is None:
                    assert k.startswith("static_initial_node") or not p.requires_grad
                    continue
                want = sum(e[k] for e in every) / world
                assert torch.allclose(p.grad, want, atol=1e-6), k
                assert not torch.equal(p.grad, mine[k]) or float(mine[k].abs().max()) == 0.0
            opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.1)
            opt.step()
            assert sh.replicas_in_sync(model.parameters()), name   # replicas stay in sync after the step
        # ranks that disagree on the layout raise together instead of entering collectives of different sizes
        c = Case("tiny_multi_uni_c2")
        torch.manual_seed(5)
        model = MultiATGCN(c.config(), c.data_feature)
        _fake_backward(model, rank, with_bucket=(rank == 0))
        with pytest.raises(RuntimeError, match="disagree"):
            sh.allreduce_model_grads_(model)
        # any nn.Module works (no gradient_exchange): all gradients through the flat copy
        lin = torch.nn.Linear(3, 2)
        with torch.no_grad():
            lin.weight.fill_(1.0); lin.bias.fill_(0.0)
        lin(torch.full((1, 3), float(rank + 1))).sum().backward()
        sh.allreduce_model_grads_(lin)
        assert torch.allclose(lin.weight.grad, torch.full((2, 3), (world + 1) / 2))
        open(os.path.join(out_dir, "dp_ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_allreduce_model_grads_two_rank_gloo(tmp_path):
    """static-feature model (bucket + leftovers), rnn_units = 32 (no bucket), both, and the plain model (bucket only):
    every gradient becomes the mean over ranks and the replicas stay bit-identical after an optimizer step"""
    world = 2
    mp.spawn(_dp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("dp_ok%d" % r)).exists() for r in range(world))


def _gpu_dp_worker(rank, world, port, out_dir):
    """one data-parallel training step per rank on the ONE GPU of the box (gloo carries the collectives): different batch
    shards, the real HIP backward, sharding.allreduce_model_grads_, Adam - the replicas must stay bit-identical"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import Case
    from multistgraph_amd.model import MultiATGCN
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        for name, has_bucket, has_rest in (("tiny_multi_uni_c2_static", True, True), ("hid32_multi_uni_c2", False, True),
                                           ("tiny_multi_uni_c2", True, False)):
            c = Case(name)
            torch.manual_seed(11)
            model = MultiATGCN(c.config("cuda:0"), c.data_feature).to(dev)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})
            model.train()
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
            rng = np.random.default_rng(100 + rank)                       # every rank its own shard
            x = torch.from_numpy(c.x + 0.1 * rng.standard_normal(c.x.shape).astype(np.float32)).to(dev)
            y = torch.from_numpy(c.y).to(dev)
            torch.manual_seed(50 + rank)                                  # and its own dropout mask
            for _ in range(2):
                opt.zero_grad()
                loss = model.calculate_loss({"X": x, "y": y})
                loss.backward()
                info = sh.allreduce_model_grads_(model)
                assert info["bucket"] == has_bucket and (info["leftover_elems"] > 0) == has_rest, (name, info)
                opt.step()
                assert sh.replicas_in_sync(model.parameters()), name
        open(os.path.join(out_dir, "gpu_dp_ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_training_steps_on_one_gpu(tmp_path):
    """the data-parallel step end to end with two ranks sharing the box's GPU over gloo: a static-feature model (bucket +
    the host-side layers' gradients), an rnn_units = 32 model (no bucket) and the plain model (bucket only)"""
    world = 2
    mp.spawn(_gpu_dp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("gpu_dp_ok%d" % r)).exists() for r in range(world))


def _rccl_one_rank_worker(rank, world, port, out_dir):
    """the same data-parallel step over the backend the product runs on - nccl = RCCL - with the one rank a one-GPU box
    allows: device-side layout vote, the flat bucket and the flat copy through RCCL all-reduces, the replica checksum on
    the GPU.  With one rank the mean is the identity: gradients must come back bit-identical."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import Case, max_norm_err
    from multistgraph_amd.model import MultiATGCN
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # what a data-parallel job does before its first forward: the path's streams in a hardware-queue pool of their own
    sh.use_own_stream_pool()
    try:
        for name, has_bucket, has_rest in (("tiny_multi_uni_c2_static", True, True), ("hid32_multi_uni_c2", False, True),
                                           ("tiny_multi_uni_c2", True, False)):
            c = Case(name)
            torch.manual_seed(11)
            model = MultiATGCN(c.config("cuda:0"), c.data_feature).to(dev)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})
            model.train()
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
            x = torch.from_numpy(c.x).to(dev)
            y = torch.from_numpy(c.y).to(dev)
            # the own-pool mode moves the work onto a library stream: the results must still be the reference's
            model.eval()
            with torch.no_grad():
                got = model.predict({"X": x}).cpu().numpy()
            if c.static_dim == 0:     # (the static-feature case draws its PCA basis at construction: no fixed prediction)
                assert max_norm_err(got, c.gold["pred"]) <= 1e-4, name
            model.train()
            for _ in range(2):
                opt.zero_grad()
                loss = model.calculate_loss({"X": x, "y": y})
                loss.backward()
                before = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
                info = sh.allreduce_model_grads_(model)
                assert info["bucket"] == has_bucket and (info["leftover_elems"] > 0) == has_rest, (name, info)
                for k, p in model.named_parameters():
                    if p.grad is not None:
                        assert torch.equal(p.grad, before[k]), (name, k)
                opt.step()
                assert sh.replicas_in_sync(model.parameters(), device=dev), name
        # the pool mode cannot be left once the streams exist; asking for the mode in use is fine
        from multistgraph_amd import _lib
        assert _lib.load().matgcn_set_stream_pool(1) == 0 and _lib.load().matgcn_set_stream_pool(0) != 0
        # the bench's own aggregate over RCCL: barrier + MAX over ranks of the elapsed time, SUM of the units
        total, slowest = sh.job_throughput(1000.0, 0.5, device=dev)
        assert abs(total - 2000.0) < 1e-6 and abs(slowest - 0.5) < 1e-9
        g = sh.gather_predictions(torch.arange(6, dtype=torch.float32, device=dev).reshape(2, 3), 2)
        assert g.shape == (2, 3)
        open(os.path.join(out_dir, "rccl_ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_gradient_exchange_over_rccl_with_the_one_rank_a_box_has(tmp_path):
    """backend "nccl" (= RCCL) on the GPU: what a one-GPU box can exercise of the N > 1 path - process-group set-up on the
    device, the layout vote and both gradient all-reduces as RCCL collectives on GPU tensors, the checksum on the GPU"""
    mp.spawn(_rccl_one_rank_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "rccl_ok0").exists()


def _mode_worker(rank, pool, out_dir):
    """one process = one stream mode (it cannot be changed once the streams exist): prediction and gradients of a golden
    case, written out for a bitwise comparison between the modes"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import Case
    from multistgraph_amd.model import MultiATGCN
    from multistgraph_amd import _lib
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if pool:
        sh.use_own_stream_pool()
    out = {}
    for name in ("tiny_multi_uni_c2", "tiny_od_non_c3"):
        c = Case(name)
        torch.manual_seed(3)
        model = MultiATGCN(c.config("cuda:0"), c.data_feature).to(dev)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})
        x = torch.from_numpy(c.x).to(dev)
        y = torch.from_numpy(c.y).to(dev)
        model.eval()
        with torch.no_grad():
            out[name + "/pred"] = model.predict({"X": x}).cpu().numpy()
        loss = model.calculate_loss({"X": x, "y": y})     # eval mode: no dropout, the training forward and the backward
        loss.backward()
        torch.cuda.synchronize()
        for k, p in model.named_parameters():
            if p.grad is not None:
                out[name + "/grad/" + k] = p.grad.detach().cpu().numpy()
    assert _lib.load().matgcn_set_stream_pool(1 if pool else 0) == 0
    np.savez(os.path.join(out_dir, "mode%d.npz" % pool), **out)


@pytest.mark.gpu
def test_own_stream_pool_gives_the_same_bits(tmp_path):
    """matgcn_set_stream_pool(1) moves the hot entry points onto a library stream and the library's streams into a queue pool
    of their own: a scheduling change only - predictions bit-identical to the default mode's, gradients equal up to the order
    of their atomic accumulations"""
    for pool in (0, 1):
        mp.spawn(_mode_worker, args=(pool, str(tmp_path)), nprocs=1, join=True)
    a, b = np.load(tmp_path / "mode0.npz"), np.load(tmp_path / "mode1.npz")
    assert sorted(a.files) == sorted(b.files) and len(a.files) > 10
    for k in a.files:
        if k.endswith("/pred"):
            assert np.array_equal(a[k], b[k]), k          # the forward: bit for bit
        else:
            # the backward accumulates some gradients with atomic adds across its streams (run-to-run differences in the
            # last bits in EITHER mode): the modes must agree to that level
            scale = float(np.abs(a[k]).max()) + 1e-30
            assert float(np.abs(a[k] - b[k]).max()) <= 2e-6 * scale, (k, float(np.abs(a[k] - b[k]).max()), scale)
