"""Per-kernel MFMA utilisation and LDS bank-conflict share of a training step from two rocprofv3 PMC passes over
serial-schedule steps (tools/pmc_train_run.sh), stamped with the source id of the build.

mfma_util         = SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) / (kernel cycles x 1024 SIMDs), kernel cycles =
                    GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs)
lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles the LDS spent re-issuing conflicting accesses
                    over the cycles it was active at all)
usage: summarise_train_pmc.py <MFMA csv> <LDS csv> <out.json> <tag>"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multistgraph_amd import build  # noqa: E402


def per_kernel(path):
    disp = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        disp[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
    agg = defaultdict(lambda: defaultdict(float))
    for (_, name), c in disp.items():
        agg[name]["launches"] += 1
        for k, v in c.items():
            agg[name][k] += v
    return agg


mfma, lds = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
out = {}
for name in sorted(mfma, key=lambda k: -mfma[k].get("GRBM_GUI_ACTIVE", 0)):
    m = mfma[name]
    if m.get("SQ_INSTS_MFMA", 0) == 0:
        continue
    n = m["launches"]
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    e = {"launches": int(n), "kernel_cycles_per_launch": cyc / n, "total_cycles": cyc,
         "mfma_instructions_per_launch": m["SQ_INSTS_MFMA"] / n,
         "mfma_util": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)}
    q = lds.get(name)
    if q and q.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        e["lds_conflict_frac"] = q["SQ_LDS_BANK_CONFLICT"] / q["SQ_LDS_IDX_ACTIVE"]
    out[name] = e
doc = {"method": __doc__.split("usage")[0].strip(), "workload": "bm403 B=64, two training steps, matgcn_set_wavefront(0)",
       "tag": sys.argv[4], "build_id": build.source_id(), "kernels": out}
json.dump(doc, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print("%-52s launches %4d  cycles/launch %9.0f  mfma_util %.3f  lds_conflict %s" % (
        k[:52], v["launches"], v["kernel_cycles_per_launch"], v["mfma_util"],
        "%.3f" % v["lds_conflict_frac"] if "lds_conflict_frac" in v else "-"))
