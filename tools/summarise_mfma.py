"""Per-kernel MFMA utilisation from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`
counter_collection.csv.  util = MFMA-busy cycles (summed over the SIMDs) / (kernel cycles * 1024 SIMDs), kernel cycles =
GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs; MI355X_MICROARCH.md, DVFS note).  usage: summarise_mfma.py <csv> <out.json>"""
import csv, json, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
disp = defaultdict(dict)
for r in rows:
    disp[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
agg = defaultdict(lambda: defaultdict(float))
for (_, name), c in disp.items():
    a = agg[name]
    a["launches"] += 1
    for k, v in c.items():
        a[k] += v
out = {}
for name, a in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    if a.get("SQ_INSTS_MFMA", 0) == 0:
        continue
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    out[name] = {"launches": int(a["launches"]), "mfma_instructions_per_launch": a["SQ_INSTS_MFMA"] / a["launches"],
                 "mfma_busy_cycles_per_launch": a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["launches"],
                 "kernel_cycles_per_launch": cyc / a["launches"],
                 "mfma_util": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)}
json.dump({"method": __doc__.split("usage")[0].strip(), "workload": "bm403 B=64, wavefront off, 2 forwards (tools/one_forward.py)",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print("%-45s launches %4d  mfma_util %.3f  busy/launch %.3e  cycles/launch %.0f" % (k[:45], v["launches"], v["mfma_util"], v["mfma_busy_cycles_per_launch"], v["kernel_cycles_per_launch"]))
