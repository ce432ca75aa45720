#!/usr/bin/env python3
"""bench.py - forward throughput of the Multi-ATGCN hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one MultiATGCN.predict() of one synthetic batch already resident in HBM.  Workload at
every N: the configuration BASELINE.json's metric is quoted on - Baltimore, 403 nodes, 24 h -> 24 h,
adjtype=multi + adpadj=unidirection (K = 5 supports), B = 64 per GPU, fp32 (exact f32 MFMA; the 1e-4
parity tolerance of the north star is an fp32 tolerance).  Batches shard over ranks with no data-path
collective (forward needs none), so scaling is weak: value = all ranks' B*T*N node-steps / max-over-ranks
time.  Parameter-only work (support stack, node-adaptive weights: matgcn_prepare) is redone inside
every timed step, as the reference redoes it in every forward.

Rank 0 prints ONE JSON line.  Two extra objects:
  roofline     - the dominant kernel (k_mix, the graph-mix GEMM): algorithmic FLOPs per launch divided by
                 its average launch duration measured with HIP events in situ, on the stream it is launched
                 on, against the dense fp32 MFMA peak.  The timed steps run the layers' chains as a
                 wavefront on several HIP streams, where a launch shares the chip with the other chain; so
                 the headline figure comes from instrumented forwards with the wavefront switched off
                 (matgcn_set_wavefront(0): same kernels, one stream - the launch duration is the kernel's
                 own), and `in_wavefront` repeats it for an instrumented forward of the timed configuration.
                 `traffic` / `mfma_util_pmc` are the HBM traffic per launch (FETCH_SIZE doubled as the gfx950
                 guide prescribes, + WRITE_SIZE) and the MFMA-busy fraction from the rocprofv3 PMC passes
                 committed under profiles/ (tools/profile_run.sh) - replayed ONLY when they were collected on the
                 build that is running (`build_id`, a hash of the library's sources), else null.
                 `node_kernels` lists the next-largest kernels (k_gate16, k_update16, k_px16) the same way.
  median       - the §8(d) protocol: 20 warm-up + 100 forwards, each bracketed by HIP events on the caller's
                 stream (the internal streams join back into it before the forward's last kernel); median,
                 p10, p90.  `value` stays the driver's contract (K steps between two barriers).
  cpu_baseline - the CPU oracle in its reference-faithful order (oracle/, kind "port"), timed on this
                 host's cores on the same workload (rank 0, N=1 only).
"""
import argparse
import datetime
import ctypes as C
import json
import os
import statistics
import sys
import time

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 matrix peak
PEAK_HBM_TBS = 8.0

WORKLOADS = {
    # name: nodes, per-GPU batch, out window, city
    "bm403": dict(nodes=403, batch=64, out=24, city="BM",
                  desc="Baltimore 403-node, in=24h out=24h, batch=64/GPU, multi+unidirection K=5"),
    "dc237": dict(nodes=237, batch=64, out=12, city="DC",
                  desc="DC 237-node, in=24h out=12h, batch=64/GPU, multi+unidirection K=5"),
    # BASELINE config 5 (stress): synthetic 4096-node graph, 32 samples per GPU (256 over 8 GPUs)
    "synth4096": dict(nodes=4096, batch=32, out=24, city="BM",
                      desc="synthetic 4096-node random .rel + learned adaptive adj, in=24h out=24h, batch=32/GPU, K=5"),
}


def algorithmic_flops_per_unit(n, k_total=5, c0=2, h=64, out=24):
    """SURVEY.md section 8d / BASELINE.md section 4."""
    total = 0.0
    for i_l in (c0 + h, 2 * h):
        total += 4 * (k_total - 1) * n * i_l + 6 * k_total * i_l * h + 6 * i_l * h
    return total + 2 * h * out


def executed_flops_per_unit(n, ks, c0=2, h=64, out=24, steps=24):
    """What the kernels really multiply, per node-step: `ks` dense supports only (diagonal ones are folded into the
    weights); the x columns are mixed once per layer, not once per AGCN; and for layers >= 1 that mix is the recurrent
    mix of the layer below (shared), except for the sequence's last step."""
    total = 0.0
    for l, c_l in enumerate((c0, h)):
        i_l = c_l + h
        x_mix = 2 * ks * n * c_l * (1.0 if l == 0 else 1.0 / steps)
        total += 4 * ks * n * h + x_mix + 6 * (ks + 1) * i_l * h + 6 * i_l * h
    return total + 2 * h * out


def backward_executed_flops(n, b, ks, k_total, c0=2, h=64, out=24, steps=24, layers=2, d=20):
    """What the kernels of matgcn_backward multiply for one batch (every product counted once, unpadded):
    chain - residual-cell transposes, the two transposed node contractions over the 1 + ks kept slots, two transposed
    graph mixes per (layer, step); x columns (node contraction of 192 columns; layer 0: its c0 channels, mixed back
    once over all steps; upper layers ride in the gate block of the layer below, only the sequence's last step is
    mixed back by itself); node-adaptive weight gradients; the adjacency gradient of the ONE learned support (the
    static supports' mixes have none); residual nn.Linear gradients; pools and node embedding; head."""
    rows = n * b
    total = 0.0
    for l in range(layers):
        c = c0 if l == 0 else h
        i_l = c + h
        per_step = (2.0 * rows * 3 * h * i_l                      # residual cell: d pre (128 + 64 columns) x W^T
                    + 2.0 * rows * h * h * (1 + ks)               # update block: h columns of the kept slots
                    + 2.0 * rows * 2 * h * h * (1 + ks)           # gate block
                    + 2 * 2.0 * ks * n * n * b * h)               # two transposed mixes
        x_cols = 2.0 * rows * 3 * h * c * (1 + ks)               # x columns of both AGCNs (192 pre-activation columns)
        x_mix = 2.0 * ks * n * n * b * c * (1.0 if l == 0 else 1.0 / steps)
        w_grad = 2.0 * rows * (1 + ks) * i_l * 3 * h             # dW[n][slot][i][o] over the step's rows
        adj = 2.0 * n * n * b * (2 * h + (c if l == 0 else h / steps))   # adaptive support: gate + update (+ own x part)
        lin = 2.0 * rows * i_l * 3 * h                            # residual nn.Linear weight gradients
        total += steps * (per_step + x_cols + x_mix + w_grad + adj + lin)
        total += 2 * 2.0 * n * d * k_total * i_l * 3 * h          # pools + node embedding from the node weight gradients
    total += 2 * 2.0 * rows * steps * h * out                     # head: weight gradient and sequence gradient
    return total


def algorithmic_bytes_per_unit(c0=2, h=64, elem=4):
    return elem * ((2 * c0 + 7 * h) + (2 * h + 7 * h))


def step_kernel_models(n, npad, b, ks, h=64):
    """FLOPs and algorithmic HBM bytes of one launch of every per-step kernel (a layer >= 1 launch: PX carries the x
    part): each operand counted once - weights streamed once per node, rows read / written once."""
    rows, kt = n * b, (1 + ks) * h
    return {
        "k_mix": dict(flops=2.0 * ks * n * n * b * h, bytes=4.0 * (ks * npad * npad + rows * h + ks * rows * h)),
        "k_gate": dict(flops=2.0 * rows * kt * 2 * h,
                       bytes=4.0 * (n * kt * 2 * h + ks * rows * h + rows * h + rows * 2 * h + rows * 2 * h)),
        "k_update": dict(flops=2.0 * rows * kt * h + 2.0 * rows * 2 * h * 3 * h,
                         bytes=4.0 * (n * kt * h + ks * rows * h + rows * h + rows * h + rows * h + rows * h + rows * h
                                      + 2 * rows * h)),
        "k_px": dict(flops=2.0 * rows * kt * 3 * h, bytes=4.0 * (n * kt * 3 * h + ks * rows * h + rows * h + rows * 3 * h)),
    }


# prefixes of the kernel names as rocprofv3 prints them (template arguments after these vary with the build)
# substrings of the profiler's kernel names (template arguments that were added later follow the ones matched here)
PMC_KERNEL_NAMES = {"k_mix": "k_mix<1", "k_gate": "k_gate16<false", "k_update": "k_update16<1, false", "k_px": "k_px16<"}
NODE_KERNEL_KEYS = {"k_gate": "k_gate16<false", "k_update": "k_update16<1, false", "k_px": "k_px16"}   # keys of roofline.node_kernels


def load_pmc(build_id):
    """the newest profiles/r*_pmc_kernels.json (tools/profile_run.sh + tools/summarise_pmc.py) that was collected on
    THIS build (`build_id`, a hash of the library's sources); PMC figures of another build are never replayed"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_kernels.json")), reverse=True)
    if not found:
        return None, "no PMC summary under profiles/"
    seen = []
    for path in found:
        with open(path) as fh:
            doc = json.load(fh)
        if doc.get("build_id") == build_id:
            return doc, "%s (build %s): %s" % (os.path.relpath(path, ROOT), build_id, doc.get("method", "").split("\n")[0])
        seen.append("%s=%s" % (os.path.basename(path), doc.get("build_id")))
    return None, "no PMC summary of build %s under profiles/ (%s): not replayed" % (build_id, ", ".join(seen[:3]))


def build_model(w, device, seed):
    from multistgraph_amd import synthetic as syn
    from multistgraph_amd.model import MultiATGCN
    df = syn.make_data_feature(w["nodes"], seed, w["city"])
    cfg = dict(input_window=24, output_window=w["out"], add_time_in_day=True, add_day_in_week=False,
               load_dynamic=False, adjtype="multi", adpadj="unidirection", cheb_order=2, embed_dim_node=20,
               embed_dim_adj=20, rnn_units=64, num_layers=2, device=device, batch_size=w["batch"])
    torch.manual_seed(seed)
    model = MultiATGCN(cfg, df).to(device).eval()   # reference init law: xavier / U(0,1)
    return model, df, cfg


def cpu_baseline(w, seed, x_np, model_state, df, pred_gpu):
    """Reference-faithful CPU oracle on the same workload, at the host's default thread count and at 8 threads (SURVEY.md
    8d: the oracle timings of the build container were taken on 8 vCPUs); ~30-60 s of CPU work in all."""
    from oracle import matgcn_oracle as O
    p = {k: v.detach().cpu() for k, v in model_state.items()}
    st = O.supports_as_tensors(O.static_supports(df["adj_mx"], df["coordinate"], None, "multi"))
    cfg = dict(adjtype="multi", adpadj="unidirection", cheb_order=2, num_layers=2, rnn_units=64,
               len_closeness=48, len_period=24, len_trend=24, output_window=w["out"], input_window=24,
               add_time_in_day=True, add_day_in_week=False, load_dynamic=False, start_dim=0, end_dim=1)
    xb = torch.from_numpy(x_np)
    units = x_np.shape[0] * 24 * w["nodes"]
    default_threads = torch.get_num_threads()

    def timed():
        with torch.no_grad():
            O.forward(xb[:4], p, st, cfg, faithful=True)          # warm-up (small)
            t0 = time.perf_counter()
            out = O.forward(xb, p, st, cfg, faithful=True)
            return out, time.perf_counter() - t0

    pred, dt = timed()
    at8 = None
    if default_threads != 8:
        torch.set_num_threads(8)
        try:
            _, dt8 = timed()
            at8 = dict(value=units / dt8, unit="node-steps/s", cores=8, seconds=dt8)
        finally:
            torch.set_num_threads(default_threads)
    err = float((pred - pred_gpu.cpu()).abs().max() / pred.abs().max())
    return dict(value=units / dt, unit="node-steps/s", cores=default_threads, kind="port",
                sample="1 reference-faithful oracle forward of the full workload (B=%d, N=%d) after a B=4 warm-up; "
                       "%.2f s on %d torch threads" % (x_np.shape[0], w["nodes"], dt, default_threads),
                seconds=dt, at_8_threads=at8, gpu_vs_cpu_max_norm_err=err), pred


def cpu_train_baseline(w, x_np, y_np, model_state, df, sample=8):
    """The training step on the host cores - calculate_loss(batch).backward() by torch autograd through the oracle
    (fp32) - on a bounded sample of the workload (`sample` of the batch's rows).  The oracle runs with the
    parameter-only work hoisted out of the time loop (faithful=False): the reference rebuilds the node-adaptive
    weights in all 96 AGCN calls, a batch-independent cost that would dominate a small sample, so this is an UPPER
    bound of the reference's CPU rate per node-step."""
    from oracle import matgcn_oracle as O
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model_state.items()}
    st = O.supports_as_tensors(O.static_supports(df["adj_mx"], df["coordinate"], None, "multi"))
    cfg = dict(adjtype="multi", adpadj="unidirection", cheb_order=2, num_layers=2, rnn_units=64,
               len_closeness=48, len_period=24, len_trend=24, output_window=w["out"], input_window=24,
               add_time_in_day=True, add_day_in_week=False, load_dynamic=False, start_dim=0, end_dim=1)
    xb, yb = torch.from_numpy(x_np[:sample]), torch.from_numpy(y_np[:sample])
    t0 = time.perf_counter()
    loss = O.calculate_loss(xb, yb, p, st, cfg, faithful=False)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    units = xb.shape[0] * 24 * w["nodes"]
    return dict(value=units / (t2 - t0), unit="node-steps/s", cores=torch.get_num_threads(), kind="port",
                sample="1 training step (calculate_loss + backward, torch autograd through the oracle with hoisted "
                       "weights: an upper bound of the reference's rate) on %d of the batch's %d rows: forward %.2f s + backward %.2f s on %d torch threads" % (
                           xb.shape[0], x_np.shape[0], t1 - t0, t2 - t1, torch.get_num_threads()),
                seconds=t2 - t0)


def in_situ_kernel_times(model, batch, wavefront=True, forwards=1):
    """Instrumented forwards: HIP events around every kernel launch of the path, on its stream."""
    from multistgraph_amd import _lib
    lib = _lib.load()
    cap = 4096 * forwards
    prev = lib.matgcn_set_wavefront(1 if wavefront else 0)
    with torch.no_grad():
        model.predict(batch)            # settle clocks / caches in this mode
    torch.cuda.synchronize()
    _lib.check(lib.matgcn_profile_enable(127, cap), "matgcn_profile_enable")
    with torch.no_grad():
        for _ in range(forwards):
            model.predict(batch)
    torch.cuda.synchronize()
    lib.matgcn_set_wavefront(prev)
    ms = (C.c_float * cap)()
    kinds = (C.c_int * cap)()
    cnt = C.c_int()
    _lib.check(lib.matgcn_profile_collect(ms, kinds, cap, C.byref(cnt)), "matgcn_profile_collect")
    lib.matgcn_profile_disable()
    out = {}
    for i in range(cnt.value):
        out.setdefault(_lib.PROF_KINDS[kinds[i]], []).append(float(ms[i]))
    return out


def train_step_times(model, batch, w, warm=2, steps=5, world=1, device=None, ctl=None, do_exchange=None):
    """One optimisation step as TrafficStateExecutor._train_epoch runs it (traffic_state_executor.py:411-422):
    loss = model.calculate_loss(batch); loss.backward(); optimizer.step() - forward_train + backward on the HIP
    path, Adam in torch.  With more than one rank every rank steps on its own batch shard and the
    gradients meet in ONE flat-bucket all-reduce (RCCL over xGMI) before the optimizer, as BASELINE config 4 asks.
    Reported beside the headline; runs last because it moves the weights."""
    from multistgraph_amd import sharding
    if do_exchange is None:
        do_exchange = world > 1    # (True with ONE rank: the RCCL rehearsal of tools/rehearse_rccl_1rank.sh)
    model.train()
    model.cache_prepared = True
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    fwd, bwd, red, adam, losses = [], [], [], [], []
    for i in range(warm + steps):
        opt.zero_grad()
        failure = None
        try:
            ev[0].record()
            loss = model.calculate_loss(batch)
            ev[1].record()
            loss.backward()
            ev[2].record()
        except Exception as exc:   # noqa: BLE001
            if not do_exchange:
                raise
            failure = exc
        if do_exchange:
            # the gradient exchange is a collective: a rank that failed its local step must not leave the others waiting
            # in it.  Every rank votes on a CPU (gloo) control group before EVERY exchange (4 bytes, negligible next to a
            # 24 ms step); one failure makes ALL ranks give the section up.
            import torch.distributed as dist
            vote = torch.tensor([0 if failure is not None else 1], dtype=torch.int32)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=ctl)
            if int(vote.item()) == 0:
                raise RuntimeError("a rank failed its local training step %d (this rank: %r)" % (i, failure))
            # ONE entry point: the flat bucket as one tensor when the gradients are views of it, plus whatever lives
            # outside it (static-feature layers; everything for rnn_units < 64)
            exchange = sharding.allreduce_model_grads_(model)
        ev[3].record()
        opt.step()
        ev[4].record()
        torch.cuda.synchronize()
        losses.append(float(loss.detach()))
        if i >= warm:
            fwd.append(ev[0].elapsed_time(ev[1])); bwd.append(ev[1].elapsed_time(ev[2]))
            red.append(ev[2].elapsed_time(ev[3])); adam.append(ev[3].elapsed_time(ev[4]))
    model.eval()
    total = statistics.mean(fwd) + statistics.mean(bwd) + statistics.mean(red) + statistics.mean(adam)
    out = {"forward_ms": statistics.mean(fwd), "backward_ms": statistics.mean(bwd),
           "grad_allreduce_ms": statistics.mean(red) if do_exchange else None, "optimizer_ms": statistics.mean(adam),
           "ms_per_step": total, "node_steps_per_s": world * w["batch"] * 24 * w["nodes"] / (total * 1e-3),
           "steps": steps, "loss_first": losses[0], "loss_last": losses[-1],
           "note": "training step through the plugin surface, rank 0's clock: HIP forward that keeps activations "
                   "(incl. the prepare after every weight update) + HIP backward behind torch autograd "
                   "(+ one flat-bucket gradient all-reduce when n_gpus > 1) + torch Adam; dropout p=0.1 on"}
    if do_exchange:
        out["grad_bucket_mb"] = sum(p.numel() for p in model.parameters() if p.requires_grad) * 4 / 1e6
        out["grad_bucket_is_one_buffer"] = bool(exchange["bucket"] and exchange["leftover_elems"] == 0)
        out["grad_exchange"] = exchange
        out["replicas_in_sync"] = sharding.replicas_in_sync(model.parameters(), device=device)
    return out


def small_batch_times(w, device, batch=16, seed=0, iters=30, steps=10):
    """The reference's SHIPPED batch size (libcity/config/model/traffic_state_pred/MultiATGCN.json:12, batch_size 16 - what
    run_model.py drives unchanged) on the workload's graph: inference forward (HIP-event median), and the training step as
    the executor runs it, three ways: device time per phase (HIP events), host time to ENQUEUE the forward and the backward
    (perf_counter around the calls, nothing synchronised inside), and the wall of a free-running loop (no synchronisation
    between steps, one at the end) - the regime in which the kernels' fixed costs and the host's enqueue rate are no longer
    hidden behind 64 samples of GPU work.  A side line, never `value`."""
    import time
    from multistgraph_amd import synthetic as syn
    ws = dict(w, batch=batch)
    model, _, _ = build_model(ws, device, seed)
    x_np, y_np = syn.make_batch_arrays(batch, w["nodes"], w["out"], seed, feat=2)
    b = {"X": torch.from_numpy(x_np).to(device), "y": torch.from_numpy(y_np).to(device)}
    with torch.no_grad():
        for _ in range(5):
            model.predict(b)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e0, e1 in evs:
            e0.record(); model.predict(b); e1.record()
        t_enq = (time.perf_counter() - t0) / iters
        torch.cuda.synchronize()
        t_wall = (time.perf_counter() - t0) / iters
    fwd = statistics.median(e0.elapsed_time(e1) for e0, e1 in evs)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    rec = {"f": [], "b": [], "o": [], "hf": [], "hb": [], "ho": []}
    for i in range(3 + steps):
        opt.zero_grad()
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        ev[0].record()
        loss = model.calculate_loss(b)
        ev[1].record()
        h1 = time.perf_counter()
        loss.backward()
        ev[2].record()
        h2 = time.perf_counter()
        opt.step()
        ev[3].record()
        h3 = time.perf_counter()
        torch.cuda.synchronize()
        if i >= 3:
            rec["f"].append(ev[0].elapsed_time(ev[1])); rec["b"].append(ev[1].elapsed_time(ev[2]))
            rec["o"].append(ev[2].elapsed_time(ev[3]))
            rec["hf"].append((h1 - h0) * 1e3); rec["hb"].append((h2 - h1) * 1e3); rec["ho"].append((h3 - h2) * 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):       # free-running: the executor's loop, no synchronisation between steps
        opt.zero_grad()
        model.calculate_loss(b).backward()
        opt.step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / steps
    m = {k: statistics.mean(v) for k, v in rec.items()}
    dev_step = m["f"] + m["b"] + m["o"]
    return {"batch": batch, "forward_ms": fwd, "forward_host_enqueue_ms": t_enq * 1e3, "forward_loop_wall_ms": t_wall * 1e3,
            "train_step_device_ms": dev_step, "train_forward_ms": m["f"], "train_backward_ms": m["b"], "optimizer_ms": m["o"],
            "host_enqueue_ms": {"forward": m["hf"], "backward": m["hb"], "optimizer": m["ho"],
                                "step": m["hf"] + m["hb"] + m["ho"]},
            "train_step_wall_ms": wall, "wall_over_device": wall / dev_step,
            "node_steps_per_s_forward": batch * 24 * w["nodes"] / (fwd * 1e-3),
            "note": "the reference's shipped batch_size (MultiATGCN.json:12); side line, never `value`: forward = HIP-event "
                    "median; training step = device time per phase, host enqueue time per phase (no synchronisation "
                    "inside), and the wall per step of a free-running loop"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="bm403", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the training-step timing (N=1 only)")
    ap.add_argument("--cache-prepared", action="store_true",
                    help="keep matgcn_prepare out of the timed steps (inference with frozen weights)")
    ap.add_argument("--no-bf16-variant", action="store_true", help="skip the bf16-operand side line")
    ap.add_argument("--no-batch16", action="store_true", help="skip the shipped-batch-size (16) side line")
    ap.add_argument("--median", type=int, default=100,
                    help="forwards of the HIP-event-timed median protocol (SURVEY.md 8d; 0 = skip)")
    ap.add_argument("--median-warmup", type=int, default=20)
    ap.add_argument("--serial-streams", action="store_true",
                    help="run the timed steps too with the layer wavefront off (one stream): the configuration the "
                         "kernel roofline is measured in; used for the rocprofv3 profile that must agree with it")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # MATGCN_BENCH_FORCE_DIST=1 (rehearsal, never set by the driver): take the multi-rank branch with ONE rank too, so
    # that a one-GPU box runs it over the real backend (nccl = RCCL): tools/rehearse_rccl_1rank.sh
    distributed = world > 1 or os.environ.get("MATGCN_BENCH_FORCE_DIST") == "1"
    # rehearsal knobs (a 2-rank dry run of the multi-rank code path on a ONE-GPU box): every rank on device 0 and
    # gloo instead of RCCL; never set by the driver
    if os.environ.get("MATGCN_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("MATGCN_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        # CPU control group (votes before optional collective sections; a dead peer raises here instead of hanging RCCL)
        ctl_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=120))

    from multistgraph_amd import build as mbuild
    from multistgraph_amd import synthetic as syn
    if rank == 0:
        mbuild.build(verbose=False)
    if distributed:
        dist.barrier()
        # inside a process group the path keeps its streams in a hardware-queue pool of their own: the communicator's streams
        # would otherwise push two of its chains onto one queue (forward 8.0 instead of 6.8 ms; sharding.use_own_stream_pool)
        if os.environ.get("MATGCN_BENCH_SHARED_POOL") != "1":      # (lab switch: the default streams)
            from multistgraph_amd import sharding as _sh
            _sh.use_own_stream_pool()

    w = dict(WORKLOADS[args.workload])
    if args.batch:
        w["batch"] = args.batch
    seed = 0
    model, df, cfg = build_model(w, device, seed)
    model.cache_prepared = bool(args.cache_prepared)
    if args.serial_streams:
        from multistgraph_amd import _lib
        _lib.load().matgcn_set_wavefront(0)
    x_np, y_np = syn.make_batch_arrays(w["batch"], w["nodes"], w["out"], seed + rank, feat=2)
    batch = {"X": torch.from_numpy(x_np).to(device), "y": torch.from_numpy(y_np).to(device)}

    def sync_all():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            pred = model.predict(batch)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pred = model.predict(batch)
        sync_all()
        elapsed = time.perf_counter() - t0
    # the same loop with matgcn_prepare cached (weights frozen between forwards - what inference serving does and
    # what the plugin class does by default); reported beside the headline, never as `value`
    frozen_elapsed = None
    if not args.cache_prepared:
        model.cache_prepared = True
        with torch.no_grad():
            model.predict(batch)
            sync_all()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pred = model.predict(batch)
            sync_all()
            frozen_elapsed = time.perf_counter() - t1
        model.cache_prepared = False
    # SURVEY.md 8d protocol: >= 20 warm-up, >= 100 forwards, every one between two HIP events, median
    median = None
    if args.median > 0:
        with torch.no_grad():
            for _ in range(args.median_warmup):
                model.predict(batch)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.median)]
            for e0, e1 in evs:
                e0.record()
                pred = model.predict(batch)
                e1.record()
            torch.cuda.synchronize()
        ts_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        med = statistics.median(ts_ms)
        median = {"forwards": args.median, "warmup": args.median_warmup, "median_ms": med,
                  "p10_ms": ts_ms[len(ts_ms) // 10], "p90_ms": ts_ms[(len(ts_ms) * 9) // 10], "min_ms": ts_ms[0],
                  "node_steps_per_s_rank0": w["batch"] * 24 * w["nodes"] / (med * 1e-3),
                  "note": "each forward bracketed by HIP events on the caller's stream, rank 0"}
    # BASELINE config 3's dtype as a side line: bf16 operands for the graph mixes (fp32 accumulate, fp32 state);
    # never `value` - narrower than the reference's fp32 - and with its own error figure against the fp32 path
    bf16_variant = None
    if rank == 0 and not args.no_bf16_variant:
        from multistgraph_amd import _lib
        lib = _lib.load()
        notes = {1: "matgcn_set_mix_precision(1): the graph mixes round their operands (support stack, state rows) to "
                    "bf16 and run on v_mfma_f32_16x16x16_bf16 with fp32 accumulation; state, node-wise contractions, "
                    "inputs and outputs stay fp32",
                 2: "matgcn_set_mix_precision(2): additionally the node-wise contractions of the recurrent step and of the "
                    "hoisted x part stream a bf16 copy of the node-adaptive weights (made once per forward, inside this "
                    "time) and round their rows to bf16 on the way into LDS; fp32 accumulation, fp32 state / residual cell"}
        with torch.no_grad():
            exact = model.predict(batch).clone()
        bf16_variant = {"tolerance": 5e-3, "note": "reported beside the f32 headline, never as `value`"}
        for mode in (1, 2):
            with torch.no_grad():
                prev_mode = lib.matgcn_set_mix_precision(mode)
                try:
                    for _ in range(max(3, args.warmup)):
                        got = model.predict(batch)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.steps):
                        got = model.predict(batch)
                    e1.record()
                    torch.cuda.synchronize()
                finally:
                    lib.matgcn_set_mix_precision(prev_mode)
            bms = e0.elapsed_time(e1) / args.steps
            entry = {"ms_per_step": bms, "node_steps_per_s_rank0": w["batch"] * 24 * w["nodes"] / (bms * 1e-3),
                     "max_norm_err_vs_f32": float((got - exact).abs().max() / exact.abs().max()), "note": notes[mode]}
            if mode == 1:
                bf16_variant.update(entry)       # the keys of round 2 keep their meaning: the mix-only variant
            else:
                bf16_variant["mix_and_node_contractions"] = entry
    units_local = w["batch"] * 24 * w["nodes"] * args.steps
    if distributed:
        from multistgraph_amd import sharding
        value, elapsed = sharding.job_throughput(units_local, elapsed, device=device)   # sum units / max time
    else:
        value = units_local / elapsed
    units_per_step = w["batch"] * 24 * w["nodes"] * world
    ms_per_step = elapsed / args.steps * 1e3

    result = None
    if rank == 0:
        flops_unit = algorithmic_flops_per_unit(w["nodes"], out=w["out"])
        fwd_tflops = flops_unit * w["batch"] * 24 * w["nodes"] / (ms_per_step * 1e-3) / 1e12
        n, b, h = w["nodes"], w["batch"], 64
        # supports the graph mix really multiplies by: diagonal ones (the similarity Laplacian is -I without
        # static features) are folded into the node-adaptive weights at prepare time and never mixed
        spec = model.spec
        n_diag = bin(spec.diag_static_mask).count("1")
        ks = (spec.n_first - n_diag) * (spec.cheb_k - 1)
        mix_flops = 2.0 * ks * n * n * b * h                       # executed = algorithmic of the dense slots, unpadded

        build_id = mbuild.source_id()
        serial = in_situ_kernel_times(model, batch, wavefront=False, forwards=2)
        conc = in_situ_kernel_times(model, batch, wavefront=True, forwards=1)
        step_mix = serial.get("k_mix", [])       # k_mix<1>: the 96 per-step launches of each forward
        mix_ms = statistics.mean(step_mix) if step_mix else float("nan")
        achieved = mix_flops / (mix_ms * 1e-3) / 1e12
        conc_mix = conc.get("k_mix", [])
        conc_ms = statistics.mean(conc_mix) if conc_mix else float("nan")
        pmc, pmc_note = (None, "PMC summary exists for bm403 B=64 only")
        if args.workload == "bm403" and w["batch"] == 64:
            pmc, pmc_note = load_pmc(build_id)

        def pmc_of(kind):
            if pmc is None:
                return None
            for name, v in pmc["kernels"].items():
                if PMC_KERNEL_NAMES[kind] in name:
                    return v
            return None

        models = step_kernel_models(n, (n + 15) // 16 * 16, b, ks)
        node_kernels = {}
        for kind in ("k_gate", "k_update", "k_px"):
            ms_list = serial.get(kind, [])
            if not ms_list:
                continue
            t = statistics.mean(ms_list) * 1e-3
            pk = pmc_of(kind)
            if kind == "k_px":   # a launch covers a chunk of 1-4 steps: scale the per-step model by the average chunk
                steps_per_launch = 24.0 * (spec.layers - 1) / (len(ms_list) / 2.0)
                models[kind] = {k: v * steps_per_launch for k, v in models[kind].items()}
            node_kernels[NODE_KERNEL_KEYS[kind]] = dict(
                launches=len(ms_list), avg_launch_ms=t * 1e3, flops_per_launch=models[kind]["flops"],
                algorithmic_bytes_per_launch=models[kind]["bytes"], achieved_tflops=models[kind]["flops"] / t / 1e12,
                frac_mfma=models[kind]["flops"] / t / 1e12 / PEAK_MFMA_F32_TFLOPS,
                achieved_algorithmic_tbs=models[kind]["bytes"] / t / 1e12,
                traffic=pk and pk["hbm_bytes_per_launch"],
                achieved_traffic_tbs=pk and pk["hbm_bytes_per_launch"] / t / 1e12,
                frac_hbm=pk and pk["hbm_bytes_per_launch"] / t / 1e12 / PEAK_HBM_TBS,
                mfma_util_pmc=pk and pk["mfma_util"])
        mix_pmc = pmc_of("k_mix")
        times = conc
        exec_unit = executed_flops_per_unit(w["nodes"], ks, out=w["out"])
        exec_tflops = exec_unit * w["batch"] * 24 * w["nodes"] / (ms_per_step * 1e-3) / 1e12
        roofline = dict(bound="mfma", kernel="k_mix<1>", achieved=achieved, peak=PEAK_MFMA_F32_TFLOPS,
                        unit="TFLOP/s", frac=achieved / PEAK_MFMA_F32_TFLOPS,
                        traffic=mix_pmc and mix_pmc["hbm_bytes_per_launch"],
                        traffic_detail=mix_pmc and dict(mix_pmc, algorithmic_bytes_per_launch=models["k_mix"]["bytes"]),
                        mfma_util_pmc=mix_pmc and mix_pmc["mfma_util"],
                        pmc_source=pmc_note,
                        launches=len(step_mix), avg_launch_ms=mix_ms,
                        flops_per_launch=mix_flops,
                        measured="HIP events around each launch, wavefront off (kernel alone on the chip)",
                        in_wavefront=dict(avg_launch_ms=conc_ms, launches=len(conc_mix),
                                          note="the per-step k_mix<1> launches of one forward of the timed configuration; "
                                               "a launch shares the chip with the other layer's chain"),
                        node_kernels=node_kernels,
                        serial_kernel_ms_per_forward={k: round(sum(v) / 2, 4) for k, v in serial.items()},
                        dense_supports_mixed=ks, supports_folded_into_weights=n_diag * (spec.cheb_k - 1),
                        whole_forward=dict(executed_tflops=exec_tflops, frac_mfma=exec_tflops / PEAK_MFMA_F32_TFLOPS,
                                           executed_flops_per_unit=exec_unit,
                                           note="executed = what the kernels multiply: dense supports only (diagonal ones "
                                                "are folded into the weights), x columns mixed once per layer and, for "
                                                "layers >= 1, shared with the recurrent mix of the layer below; frac_mfma "
                                                "= executed / the dense fp32 MFMA peak",
                                           survey_formula_tflops=fwd_tflops, survey_formula_flops_per_unit=flops_unit,
                                           survey_formula_note="SURVEY section 8d formula / time: counts all K-1 supports "
                                                               "as dense and the x columns in both AGCNs of every layer, "
                                                               "i.e. FLOPs these kernels no longer execute - NOT a "
                                                               "utilisation figure",
                                           algorithmic_gbs=algorithmic_bytes_per_unit() * units_per_step / world /
                                           (ms_per_step * 1e-3) / 1e9),
                        kernel_ms_per_forward={k: round(sum(v), 4) for k, v in times.items()})
        ytrue = batch["y"][..., 0:1].clone()
        from multistgraph_amd.model import masked_mae
        mae12 = float(masked_mae(pred[:, min(12, w["out"]) - 1], ytrue[:, min(12, w["out"]) - 1]).item())
        result = {
            "metric": "forward node-steps/s (B*T*N per second), MultiATGCN.predict",
            "value": value, "unit": "node-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["desc"], "nodes": w["nodes"], "per_gpu_batch": w["batch"],
                       "global_batch": w["batch"] * world, "in_steps": 24, "out_steps": w["out"],
                       "parallelism": "batch shards, no data-path collective (dp%d)" % world,
                       "prepare_in_timed_region": not args.cache_prepared,
                       "layer_wavefront_streams": not args.serial_streams},
            "mae_at_12": mae12,
            "build_id": build_id,
            "roofline": roofline,
        }
        if median is not None:
            result["median"] = median
        if bf16_variant is not None:
            result["bf16_variant"] = bf16_variant
        if frozen_elapsed is not None:
            result["frozen_weights"] = {
                "ms_per_step": frozen_elapsed / args.steps * 1e3,
                "value_rank0": w["batch"] * 24 * w["nodes"] * args.steps / frozen_elapsed,
                "note": "same steps with the parameter-only work (matgcn_prepare) cached between forwards; "
                        "rank 0's own rate, not part of `value`"}
        # Order (round 4): every GPU side line first, on a quiet host, the CPU baselines LAST.  The oracle's 128 torch
        # threads keep spinning for a while after their last parallel region; the training step and the B = 16 side line
        # that used to follow them measured that (host enqueue 6.1 instead of 3.6 ms per B = 16 step, the training forward
        # 0.3-0.7 ms slower than tools/host_enqueue_time.py on the same box).  The CPU runs take the parameters as they
        # were for the timed forwards: a snapshot from before the training steps move them.
        want_cpu = world == 1 and not args.no_cpu_baseline and args.workload != "synth4096"   # (hours of CPU time at N=4096)
        snapshot = {k: v.detach().cpu().clone() for k, v in model.named_parameters()} if want_cpu else None
        pred_fp32 = pred.detach().cpu().clone() if want_cpu else None   # (the output buffer is reused by later forwards)
        if world == 1 and not distributed and not args.no_train_step:
            ts = train_step_times(model, batch, w)
            # every forward GEMM has two backward GEMMs (input gradient, weight gradient): the backward executes ~2x the
            # forward's EXECUTED FLOPs (dense supports only, shared mixes) - not 2x the SURVEY formula
            bwd_flops = backward_executed_flops(w["nodes"], w["batch"], ks, spec.k_total, out=w["out"])
            ts["backward_executed_gflop"] = bwd_flops / 1e9
            ts["backward_executed_tflops"] = bwd_flops / (ts["backward_ms"] * 1e-3) / 1e12
            ts["backward_frac_mfma"] = ts["backward_executed_tflops"] / PEAK_MFMA_F32_TFLOPS
            ts["backward_over_forward_flops"] = bwd_flops / (exec_unit * w["batch"] * 24 * w["nodes"])
            ts["backward_flops_note"] = ("executed FLOPs of matgcn_backward by the kernel models of "
                                         "bench.backward_executed_flops (every product once, unpadded; only the learned "
                                         "support has an adjacency-gradient GEMM) / backward time: a utilisation of the "
                                         "dense fp32 MFMA peak")
            result["train_step"] = ts
        if world == 1 and not args.no_batch16 and args.workload != "synth4096" and w["batch"] != 16:
            # the reference's shipped batch size (MultiATGCN.json:12) on this workload's graph - and on the other headline
            # graph when this is the default workload: forward, training step, host enqueue time.  Never `value`.
            try:
                b16 = {args.workload: small_batch_times(WORKLOADS[args.workload], device)}
                if args.workload == "bm403":
                    b16["dc237"] = small_batch_times(WORKLOADS["dc237"], device)
                result["batch16"] = b16
            except Exception as exc:   # noqa: BLE001 - a side line must not take the headline with it
                result["batch16"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if want_cpu:
            base, pred_cpu = cpu_baseline(w, seed, x_np, snapshot, df, pred_fp32)
            result["cpu_baseline"] = base
            result["mae_at_12_cpu"] = float(masked_mae(pred_cpu[:, min(12, w["out"]) - 1],
                                                       torch.from_numpy(y_np)[:, min(12, w["out"]) - 1, :, 0:1]).item())
            best_cpu = max(base["value"], (base.get("at_8_threads") or {}).get("value", 0.0))
            result["gpu_over_cpu"] = value / best_cpu       # against the FASTER of the two CPU runs
            result["gpu_over_cpu_basis"] = "the faster CPU run: %d threads" % (
                base["cores"] if best_cpu == base["value"] else 8)
            if "train_step" in result and "error" not in result["train_step"]:
                cpu_train = cpu_train_baseline(w, x_np, y_np, snapshot, df)
                result["train_step"]["cpu_baseline"] = cpu_train
                result["train_step"]["gpu_over_cpu"] = result["train_step"]["node_steps_per_s"] / cpu_train["value"]
    if distributed and not args.no_train_step:
        # every rank takes part (the gradient all-reduce is a collective); rank 0 reports
        try:
            ts = train_step_times(model, batch, w, world=world, device=device, ctl=ctl_group, do_exchange=True)
        except Exception as exc:   # noqa: BLE001 - the headline line must survive a failure of the optional section
            ts = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if rank == 0:
            result["train_step"] = ts
    if rank == 0:
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
