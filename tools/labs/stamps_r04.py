#!/usr/bin/env python3
"""Round 4 lab: where does the time of a k_gate16 / k_update16 launch go?

Needs the lab build lib/libmatgcn_stamps.so (-DNODE_LAB_STAMPS: every wave keeps s_memtime stamps of its phases in
SGPRs and writes them out at its end, with its HW_ID / XCC_ID).  Runs ONE forward of a bench workload on the serial
schedule (every kernel alone on the chip) inside the product's own data flow, reads the stamps back and prints, per
kernel, the phase durations of a workgroup - separately for workgroups that had a CU to themselves and for those that
shared one - and the launch's span.

    MATGCN_LIB=multistgraph_amd/lib/libmatgcn_stamps.so python tools/labs/stamps_r04.py [--workload bm403] [--wavefront]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NS = 28
GATE_NAMES = {0: "entry", 1: "requests issued", 2: "s rows -> LDS", 3: "barrier", 4: "chunk0 issued", 5: "chunks1,2 -> LDS",
              6: "barrier", 7: "odd chunk issued", 8: "barrier", 9: "even chunk issued", 10: "barrier", 11: "late requests",
              12: "last chunk issued", 13: "acc + PX", 14: "barrier", 15: "sigmoid, tile->LDS", 16: "barrier",
              17: "stores issued", 18: "stores acked"}
UPD_NAMES = {0: "entry", 1: "requests issued", 2: "s rows -> LDS", 3: "barrier", 4: "chunk0 issued", 5: "chunks1,2 -> LDS",
             6: "barrier", 7: "odd chunk issued", 8: "barrier", 9: "even chunk issued", 10: "barrier", 11: "late requests",
             12: "last chunk issued", 13: "acc + PX", 14: "barrier", 15: "tanh, h', x_t -> LDS", 16: "barrier",
             17: "res GEMM1 + sigmoid", 18: "barrier", 19: "res GEMM2 issued", 20: "tanh + blend (after barrier)",
             21: "stores acked (after barrier)"}


def analyse(name, st, names, clock_ghz_out):
    """st: [blocks][8 waves][NS] uint32"""
    blocks = st.shape[0]
    live = st[:, 0, 0] != 0
    st = st[live]
    hw = st[:, 0, 26]
    xcc = st[:, 0, 27] & 0xF
    cu = ((hw >> 8) & 0xFF).astype(np.int64) + 256 * xcc.astype(np.int64)
    uniq, inv, cnt = np.unique(cu, return_inverse=True, return_counts=True)
    per_cu = cnt[inv]
    t = st[:, :, :24].astype(np.int64)
    order = [k for k in sorted(names) if k < 24]
    # wave 0's view and the slowest wave's view of each stamp, relative to the workgroup's entry
    t0 = t[:, :, 0].min(axis=1)
    rel = (t - t0[:, None, None])
    rel[t == 0] = -1
    end = rel[:, :, order[-1]].max(axis=1)
    print("== %s: %d workgroups live of %d, on %d CUs (%d CUs with 2, %d with 1, %d with >2)" % (
        name, len(st), blocks, len(uniq), int((cnt == 2).sum()), int((cnt == 1).sum()), int((cnt > 2).sum())))
    # launch span from the 100 MHz real-time counter at flush (end of each wave) and entry offsets via s_memtime of the
    # same XCD are not comparable across XCDs; so: span = (max realtime at end - min realtime at end) + median duration
    rt0 = st[:, :, 24].astype(np.int64)
    rt1 = st[:, :, 25].astype(np.int64)
    dur_us = (rt1.max(axis=1) - rt0.min(axis=1)) / 100.0
    ghz = np.median(end / np.maximum(dur_us, 1e-3)) / 1e3
    print("   workgroup duration: median %d cycles = %.2f us (s_memtime tick rate %.3f GHz)  p10 %d  p90 %d  max %d cycles" % (
        np.median(end), np.median(dur_us), ghz, np.percentile(end, 10), np.percentile(end, 90), end.max()))
    print("   launch: first entry -> last end %.2f us; entries spread over %.2f us; ends spread over %.2f us" % (
        (rt1.max() - rt0.min()) / 100.0, (rt0.min(axis=1).max() - rt0.min()) / 100.0, (rt1.max() - rt1.max(axis=1).min()) / 100.0))
    for label, sel in (("alone on its CU", per_cu == 1), ("sharing a CU", per_cu >= 2)):
        if not sel.any():
            continue
        r = rel[sel]
        prev = None
        line = []
        for k in order:
            v = r[:, :, k]
            v = np.where(v < 0, np.nan, v)
            med = np.nanmedian(np.nanmax(v, axis=1))   # slowest wave reaches stamp k
            if np.isnan(med):
                continue
            line.append((k, names[k], med, None if prev is None else med - prev))
            prev = med
        print("   -- %s (%d workgroups): stamp | cycles since entry (slowest wave, median over workgroups) | delta" % (
            label, int(sel.sum())))
        for k, nm, med, d in line:
            print("      %2d %-30s %8.0f %s" % (k, nm, med, "" if d is None else "%+8.0f" % d))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="bm403")
    ap.add_argument("--wavefront", action="store_true")
    ap.add_argument("--launch", type=int, default=20, help="which (gate, update) pair of the forward to print")
    args = ap.parse_args()
    from multistgraph_amd import _lib, synthetic as syn
    dev = torch.device("cuda", 0)
    w = dict(bench.WORKLOADS[args.workload])
    model, df, cfg = bench.build_model(w, dev, 0)
    lib = _lib.load()
    lib.matgcn_lab_stamps.restype = C.c_int
    lib.matgcn_lab_stamps.argtypes = [C.c_void_p, C.c_size_t]
    if not args.wavefront:
        lib.matgcn_set_wavefront(0)
    x_np, y_np = syn.make_batch_arrays(w["batch"], w["nodes"], w["out"], 0, feat=2)
    batch = {"X": torch.from_numpy(x_np).to(dev)}
    with torch.no_grad():
        for _ in range(5):
            model.predict(batch)
        torch.cuda.synchronize()
        blocks = (w["nodes"] + 7) // 8 * 8 * ((w["batch"] + 63) // 64)
        per = blocks * 8 * NS
        launches = 2 * 2 * 24
        buf = torch.zeros(per * launches, dtype=torch.int32, device=dev)
        lib.matgcn_lab_stamps(buf.data_ptr(), buf.numel())
        # clock: s_memtime ticks against wall time
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); model.predict(batch); e1.record()
        torch.cuda.synchronize()
        n = lib.matgcn_lab_stamp_launches()
        lib.matgcn_lab_stamps(None, 0)
    print("forward with stamps: %.3f ms, %d stamped launches" % (e0.elapsed_time(e1), n))
    st = buf.cpu().numpy().view(np.uint32).reshape(launches, blocks, 8, NS)
    k = args.launch
    for idx, nm, names in ((2 * k, "k_gate16 (layer %d, step %d)" % (k // 24, k % 24), GATE_NAMES),
                           (2 * k + 1, "k_update16<1> (layer %d, step %d)" % (k // 24, k % 24), UPD_NAMES)):
        analyse(nm, st[idx], names, None)
    # durations over all launches of a kind (median workgroup duration), to see whether the printed one is typical
    for kind, off in (("gate", 0), ("update", 1)):
        meds = []
        for i in range(off, n, 2):
            s = st[i]
            live = s[:, 0, 0] != 0
            t = s[live][:, :, :24].astype(np.int64)
            last = 18 if kind == "gate" else 21
            meds.append(np.median((t[:, :, last].max(axis=1) - t[:, :, 0].min(axis=1))))
        print("%s: median workgroup duration over %d launches: min %d  median %d  max %d cycles" % (
            kind, len(meds), min(meds), np.median(meds), max(meds)))


if __name__ == "__main__":
    main()
