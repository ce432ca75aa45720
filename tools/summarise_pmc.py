"""Per-kernel HBM traffic and MFMA utilisation of the step kernels from three rocprofv3 PMC passes over two
serial-schedule forwards (tools/profile_run.sh, tools/one_forward.py), stamped with the source id of the build.

traffic  = 2 x FETCH_SIZE + WRITE_SIZE per launch (units KB; FETCH_SIZE doubled: gfx950 tallies the 128-byte reads of a
           wide coalesced stream at 64 B - MI355X_MICROARCH.md, HBM section; the two counters need separate passes)
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) / (kernel cycles x 1024 SIMDs), kernel cycles =
           GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs)
usage: summarise_pmc.py <FETCH csv> <WRITE csv> <MFMA csv> <out.json> <tag>"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multistgraph_amd import build  # noqa: E402


def per_kernel(path, counters):
    disp = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in counters:
            disp[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
    agg = defaultdict(lambda: defaultdict(float))
    for (_, name), c in disp.items():
        agg[name]["launches"] += 1
        for k, v in c.items():
            agg[name][k] += v
    return agg


fetch = per_kernel(sys.argv[1], {"FETCH_SIZE"})
write = per_kernel(sys.argv[2], {"WRITE_SIZE"})
mfma = per_kernel(sys.argv[3], {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE"})
out = {}
for name in sorted(mfma, key=lambda k: -mfma[k].get("GRBM_GUI_ACTIVE", 0)):
    m = mfma[name]
    if m.get("SQ_INSTS_MFMA", 0) == 0 or name not in fetch or name not in write:
        continue
    n = m["launches"]
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    f_kb = fetch[name]["FETCH_SIZE"] / fetch[name]["launches"]
    w_kb = write[name]["WRITE_SIZE"] / write[name]["launches"]
    out[name] = {"launches": int(n), "FETCH_SIZE_KB_avg": f_kb, "WRITE_SIZE_KB_avg": w_kb,
                 "read_bytes_per_launch": 2.0 * f_kb * 1024, "write_bytes_per_launch": w_kb * 1024,
                 "hbm_bytes_per_launch": 2.0 * f_kb * 1024 + w_kb * 1024,
                 "mfma_instructions_per_launch": m["SQ_INSTS_MFMA"] / n,
                 "mfma_busy_cycles_per_launch": m["SQ_VALU_MFMA_BUSY_CYCLES"] / n,
                 "kernel_cycles_per_launch": cyc / n, "mfma_util": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)}
doc = {"method": __doc__.split("usage")[0].strip(), "workload": "bm403 B=64, wavefront off, 2 forwards (tools/one_forward.py)",
       "tag": sys.argv[5], "build_id": build.source_id(), "kernels": out}
json.dump(doc, open(sys.argv[4], "w"), indent=1)
for k, v in out.items():
    print("%-44s launches %4d  HBM %7.1f MB/launch  mfma_util %.3f  cycles/launch %.0f" % (
        k[:44], v["launches"], v["hbm_bytes_per_launch"] / 1e6, v["mfma_util"], v["kernel_cycles_per_launch"]))
