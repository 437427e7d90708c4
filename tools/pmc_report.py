"""Summarise the two SQ PMC passes of tools/pmc_raster.sh per raster kernel (per-mesh figures, N = argv[2])."""
import collections
import csv
import glob
import json
import sys

out, n_mesh = sys.argv[1], int(sys.argv[2])
d = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "raster" in r["Kernel_Name"]:
            d[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    c = {name: sum(x) / len(x) for name, x in v.items()}
    dur = c.get("GRBM_GUI_ACTIVE", 0) / 8.0                   # summed over the 8 XCDs
    rep = {"kernel": k, "cycles": round(dur), "valu_insts_per_mesh": round(c.get("SQ_INSTS_VALU", 0) / n_mesh),
           "salu_insts_per_mesh": round(c.get("SQ_INSTS_SALU", 0) / n_mesh),
           "lds_insts_per_mesh": round(c.get("SQ_INSTS_LDS", 0) / n_mesh),
           "vmem_rd_per_mesh": round(c.get("SQ_INSTS_VMEM_RD", 0) / n_mesh), "waves": c.get("SQ_WAVES")}
    if dur:
        rep["avg_waves_per_simd"] = round(c.get("SQ_WAVE_CYCLES", 0) * 4 / (dur * 1024), 2)      # quad-cycle units
        rep["valu_busy_frac"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (dur * 1024), 3)
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        rep["wave_wait_frac"] = round(c.get("SQ_WAIT_ANY", 0) / wc, 3)
        rep["wave_issue_stall_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
        rep["wave_lds_stall_frac"] = round(c.get("SQ_WAIT_INST_LDS", 0) / wc, 3)
        rep["wave_active_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
    if c.get("SQ_ACTIVE_INST_VALU"):
        rep["lane_activity"] = round(c.get("SQ_THREAD_CYCLES_VALU", 0) / (c["SQ_ACTIVE_INST_VALU"] * 64), 3)
    print(json.dumps(rep))
