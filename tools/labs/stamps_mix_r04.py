#!/usr/bin/env python3
"""Round 4 lab: where does the time of a k_mix<1> launch go?  (lib/libmatgcn_stamps.so, serial schedule.)
Per wave: entry, requests issued, A / B tile 0 arrived, first barrier, K loop issued, accumulators in LDS, stores issued,
stores acknowledged; plus the launch span (100 MHz real-time counter) and the spread of entries / ends over the grid.
    MATGCN_LIB=multistgraph_amd/lib/libmatgcn_stamps.so python tools/labs/stamps_mix_r04.py [--workload bm403] [--launch 40]"""
import argparse, ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
NS = 28
NAMES = ["entry", "six requests issued", "A tile 0 arrived", "B tile 0 arrived", "first barrier", "K loop issued",
         "accumulators -> LDS", "stores issued", "stores acknowledged"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="bm403")
    ap.add_argument("--launch", type=int, default=40)
    args = ap.parse_args()
    from multistgraph_amd import _lib, synthetic as syn
    dev = torch.device("cuda", 0)
    w = dict(bench.WORKLOADS[args.workload])
    model, df, cfg = bench.build_model(w, dev, 0)
    lib = _lib.load()
    lib.matgcn_lab_stamps.argtypes = [C.c_void_p, C.c_size_t]
    lib.matgcn_set_wavefront(0)
    lib.matgcn_lab_stamp_kinds(2)
    x_np, _ = syn.make_batch_arrays(w["batch"], w["nodes"], w["out"], 0, feat=2)
    batch = {"X": torch.from_numpy(x_np).to(dev)}
    npad = (w["nodes"] + 15) // 16 * 16
    ks = 3
    blocks = ((ks * npad + 63) // 64) * w["batch"]
    per = blocks * 4 * NS
    launches = 96
    with torch.no_grad():
        for _ in range(5):
            model.predict(batch)
        torch.cuda.synchronize()
        buf = torch.zeros(per * launches, dtype=torch.int32, device=dev)
        lib.matgcn_lab_stamps(buf.data_ptr(), buf.numel())
        model.predict(batch)
        torch.cuda.synchronize()
        n = lib.matgcn_lab_stamp_launches()
        lib.matgcn_lab_stamps(None, 0)
    st = buf.cpu().numpy().view(np.uint32).reshape(launches, blocks, 4, NS)
    print("%d stamped k_mix<1> launches, %d workgroups each" % (n, blocks))
    s = st[args.launch]
    live = s[:, 0, 0] != 0
    s = s[live]
    t = s[:, :, :9].astype(np.int64)
    t0 = t[:, :, 0].min(axis=1)
    rel = t - t0[:, None, None]
    rt0, rt1 = s[:, :, 24].astype(np.int64), s[:, :, 25].astype(np.int64)
    dur_us = (rt1.max(axis=1) - rt0.min(axis=1)) / 100.0
    end = rel[:, :, 8].max(axis=1)
    print("launch %d: %d workgroups; duration median %d cycles = %.2f us (%.3f GHz); p10 %d p90 %d max %d" % (
        args.launch, len(s), np.median(end), np.median(dur_us), np.median(end / np.maximum(dur_us, 1e-3)) / 1e3,
        np.percentile(end, 10), np.percentile(end, 90), end.max()))
    print("launch span: first entry -> last end %.2f us; entries spread over %.2f us; ends spread over %.2f us" % (
        (rt1.max() - rt0.min()) / 100.0, (rt0.min(axis=1).max() - rt0.min()) / 100.0, (rt1.max() - rt1.max(axis=1).min()) / 100.0))
    prev = None
    for k, nm in enumerate(NAMES):
        med = np.median(rel[:, :, k].max(axis=1))
        print("   %d %-24s %8.0f %s" % (k, nm, med, "" if prev is None else "%+8.0f" % (med - prev)))
        prev = med
    hw = s[:, 0, 26]; xcc = s[:, 0, 27] & 0xF
    cu = ((hw >> 8) & 0xFF).astype(np.int64) + 256 * xcc.astype(np.int64)
    uniq, cnt = np.unique(cu, return_counts=True)
    print("   CUs used %d; workgroups per CU: min %d median %d max %d" % (len(uniq), cnt.min(), np.median(cnt), cnt.max()))
    # per XCD: workgroup duration (ticks and us), K-loop ticks, and when its workgroups entered / ended relative to the launch
    kl = rel[:, :, 5].max(axis=1) - rel[:, :, 4].max(axis=1)
    base = rt0.min()
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print("   XCD %d: %4d workgroups  duration median %6d ticks = %5.2f us  K loop %6d ticks  entry %5.2f us  end median %5.2f max %5.2f us" % (
            x, int(m.sum()), np.median(end[m]), np.median(dur_us[m]), np.median(kl[m]),
            (np.median(rt0[m].min(axis=1)) - base) / 100.0, (np.median(rt1[m].max(axis=1)) - base) / 100.0,
            (rt1[m].max() - base) / 100.0))
    # the CUs of one XCD: is the spread between CUs or inside them?
    x0 = xcc == sorted(set(xcc.tolist()))[0]
    cu0 = cu[x0]
    byc = [(c, np.median(end[x0][cu0 == c]), end[x0][cu0 == c].min(), end[x0][cu0 == c].max()) for c in sorted(set(cu0.tolist()))]
    byc.sort(key=lambda r: r[1])
    print("   XCD %d per CU (median / min / max workgroup ticks), fastest and slowest 4:" % sorted(set(xcc.tolist()))[0])
    for r in byc[:4] + byc[-4:]:
        print("      CU %4d  %6d  %6d  %6d" % r)


if __name__ == "__main__":
    main()
