"""Batch sharding of the hot path over the GPUs of one node.

The forward path shards on the batch axis only: parameters, supports and the ``prepared`` buffer are
replicated, every rank runs its own slice of the global batch, and the forward needs **no collective**.
``torch.distributed`` (backend ``nccl`` = RCCL on ROCm, ``gloo`` in the CPU tests) is used for the barrier /
max-over-ranks timing of ``bench.py`` and for the one exchange a training step adds: a flat-bucket gradient
all-reduce (15.5 MB fp32 at N=403 - one bucket, because xGMI rings are per-link bound and a single large
message amortises the ring latency best).

Nothing here touches the device path; it is plain host logic and is covered by world_size-2 gloo tests.
"""
from __future__ import annotations

from typing import Iterable, Sequence, Tuple

import torch
import torch.distributed as dist


def use_own_stream_pool() -> None:
    """For data-parallel jobs: call once per process BEFORE the first forward of the HIP path.  The library then keeps its
    streams in a hardware-queue pool of their own (highest stream priority) and runs its hot entry points on a library
    stream instead of the caller's, so that an RCCL communicator's streams cannot push two of the path's chains onto one
    hardware queue (forward 8.0 instead of 6.8 ms at the headline shape; ``matgcn_set_stream_pool`` in include/matgcn.h,
    profiles/r04_rccl_queues_lab.log).  Raises if the streams already exist in the other mode."""
    from . import _lib
    _lib.check(_lib.load().matgcn_set_stream_pool(1), "matgcn_set_stream_pool")


def shard_bounds(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) rows of the global batch owned by ``rank``: contiguous, sizes differ by at most one
    (the first ``global_batch % world`` ranks take the extra row)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank %d of world %d" % (rank, world))
    if global_batch < 0:
        raise ValueError("negative batch")
    base, extra = divmod(global_batch, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Slice every tensor of a LibCity-style batch dict (``X``, ``y``) on axis 0."""
    some = next(iter(batch.values()))
    lo, hi = shard_bounds(int(some.shape[0]), rank, world)
    return {k: v[lo:hi] for k, v in batch.items()}


def gather_predictions(local: torch.Tensor, global_batch: int, group=None) -> torch.Tensor:
    """Concatenate the ranks' predictions in rank order (evaluation only; ragged shards allowed)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(global_batch, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    assert sizes[rank][1] - sizes[rank][0] == local.shape[0]
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], 0)


def job_throughput(units_local: float, seconds_local: float, device=None, group=None) -> Tuple[float, float]:
    """Whole-job rate of a weak/strong-scaled run: sum of the ranks' units over the slowest rank's time
    (the bench.py contract).  Returns (units_per_second, max_seconds)."""
    t = torch.tensor([seconds_local, units_local], dtype=torch.float64, device=device)
    tmax = t[:1].clone()
    usum = t[1:].clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(usum, op=dist.ReduceOp.SUM, group=group)
    return float(usum.item() / tmax.item()), float(tmax.item())


def flat_allreduce_mean_(tensors: Sequence[torch.Tensor], group=None) -> None:
    """In-place mean over ranks of a list of tensors through ONE flat bucket: the gradient exchange of
    data-parallel training, after matgcn_backward has filled the ranks' gradients."""
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def bucket_allreduce_mean_(bucket: torch.Tensor, group=None) -> None:
    """In-place mean over ranks of the flat gradient buffer itself (HotPath.grad_bucket / MultiATGCN.gradient_bucket):
    the parameters' ``.grad`` are views of it, so this ONE collective on ONE tensor is the whole gradient exchange - no
    concatenation, no copy back."""
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    bucket /= dist.get_world_size(group)


def allreduce_model_grads_(model, group=None, check: bool = True) -> dict:
    """THE gradient exchange of data-parallel training: in-place mean over ranks of every gradient of ``model`` after
    ``loss.backward()``.  The flat bucket of the HIP path is reduced as ONE tensor when the model's gradients are
    views of it (``model.gradient_exchange()``); whatever lives outside it - the host-side ``static_initial_*``
    layers of the static-feature path, or every gradient when there is no bucket (rnn_units < 64, accumulated
    gradients) - goes through one flat copy.  Works for any nn.Module (no ``gradient_exchange``: all gradients
    through the flat copy).

    ``check`` (default): the ranks first agree on the layout with one 3-number all-reduce (MIN and MAX of
    [has bucket, bucket elements, leftover elements]); a rank whose layout differs would otherwise enter collectives
    of other sizes than its peers and hang or corrupt them - here every rank raises instead.
    Returns {"bucket": bool, "bucket_elems": int, "leftover_elems": int}."""
    if hasattr(model, "gradient_exchange"):
        bucket, rest = model.gradient_exchange()
    else:
        bucket, rest = None, [p.grad for p in model.parameters() if p.requires_grad and p.grad is not None]
    layout = [0 if bucket is None else 1, 0 if bucket is None else bucket.numel(), sum(t.numel() for t in rest)]
    if check:
        # where the layout vote travels: under nccl / RCCL every rank needs a GPU tensor - also a rank that has no
        # gradient at all (it used to build a CPU tensor, raise alone and leave its peers in the
        # collective): take the device from the gradients, else from the parameters, else the current device
        dev = None
        if dist.get_backend(group) == "nccl":
            some = bucket if bucket is not None else (rest[0] if rest else None)
            if some is None:
                some = next((p for p in model.parameters() if p.is_cuda), None) if hasattr(model, "parameters") else None
            dev = some.device if some is not None and some.is_cuda else torch.device("cuda", torch.cuda.current_device())
        lo = torch.tensor(layout, dtype=torch.int64, device=dev)
        both = torch.stack([lo, -lo])                 # one MIN gives the minimum and (negated) the maximum
        dist.all_reduce(both, op=dist.ReduceOp.MIN, group=group)
        if not torch.equal(both[0], -both[1]):
            raise RuntimeError("allreduce_model_grads_: the ranks disagree on the gradient layout (this rank: bucket=%d, "
                               "%d bucket elements, %d leftover elements; min over ranks %s, max %s)" % (
                                   layout[0], layout[1], layout[2], both[0].tolist(), (-both[1]).tolist()))
    if bucket is not None:
        bucket_allreduce_mean_(bucket, group=group)
    flat_allreduce_mean_(rest, group=group)
    return {"bucket": bucket is not None, "bucket_elems": layout[1], "leftover_elems": layout[2]}


def replicas_in_sync(params: Iterable[torch.Tensor], group=None, device=None) -> bool:
    """True when every rank holds bit-identical parameters (checksum all-reduce MIN/MAX).  ``device``: where the
    two checksums travel (a GPU for the nccl/RCCL backend, None = CPU for gloo)."""
    acc = torch.zeros(2, dtype=torch.float64)
    for p in params:
        v = p.detach().double().cpu()
        acc[0] += v.sum()
        acc[1] += (v * v).sum()
    if device is not None:
        acc = acc.to(device)
    lo, hi = acc.clone(), acc.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))
