"""GPU box: rocprofv3 PMC passes (one counter group per run, --kernel-trace only) over tools/kernels.py -- or, with
PMC_TARGET="tools/scene_times.py profiles/scenes/live_s1_a.npz --iters 3", over any other script of this tree --, restricted to
the counters `rocprofv3 -L` lists on this box, and a per-kernel summary (incl. VALU busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x
kernel cycles at 2.4 GHz), from the same pass's kernel trace).  usage: pmc_passes.py <outdir> [kernels.py args...]"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1])
extra = sys.argv[2:]
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
avail_txt = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, cwd="/tmp", env=env)
open(os.path.join(out, "counters_list.txt"), "w").write(avail_txt.stdout + avail_txt.stderr)
avail = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", avail_txt.stdout + avail_txt.stderr))
GROUPS = {
    "sq1": "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS",
    "sq2": "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS",
    "sq3": "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM SQ_WAVE_CYCLES",
    "sq4": "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_BRANCH",
    "sqc": "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_INPUT_VALID_READYB SQC_TC_REQ",
    "tcc1": "TCC_HIT_sum TCC_MISS_sum",
    "tcc2": "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum",
    "tcp": "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum",
}
used = {}
only = os.environ.get("PMC_GROUPS")
for g, names in GROUPS.items():
    if only and g not in only.split(","):
        continue
    ok = [n for n in names.split() if n in avail]
    if not ok:
        continue
    used[g] = ok
    target = os.environ.get("PMC_TARGET", "").split()
    tail = ([os.path.join(R, target[0])] + [os.path.join(R, a) if a.endswith(".npz") else a for a in target[1:]]) if target else \
        ([os.path.join(R, "tools/kernels.py")] + (extra or ["3"]))
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ok + ["--output-format", "csv", "-d", os.path.join(out, g), "-o", "t", "--", sys.executable] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=env)
    open(os.path.join(out, g + ".log"), "w").write(r.stdout[-4000:] + "\n---\n" + r.stderr[-4000:])
d = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if "k_raster" in k or "k_face" in k or "k_superblock" in k:
            d[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)          # kernel durations of the sq2 pass (the one that holds SQ_ACTIVE_INST_VALU)
for fn in glob.glob(out + "/sq2/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        dur[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rep = {"counters_used": used, "target": os.environ.get("PMC_TARGET", "tools/kernels.py"), "kernels": {}}
for k, v in sorted(d.items()):
    c = {n: sum(x) / len(x) for n, x in v.items()}
    c["launches_seen"] = max(len(x) for x in v.values())
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_SMEM",
                  "SQ_INST_LEVEL_LDS"):
            if n in c:
                c[n + "/WAVE_CYCLES"] = round(c[n] / wc, 4)
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        c["tcc_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if c.get("SQC_DCACHE_HITS") is not None and c.get("SQC_DCACHE_MISSES") is not None and c["SQC_DCACHE_HITS"] + c["SQC_DCACHE_MISSES"] > 0:
        c["scalar_cache_hit_rate"] = round(c["SQC_DCACHE_HITS"] / (c["SQC_DCACHE_HITS"] + c["SQC_DCACHE_MISSES"]), 4)
    if c.get("SQ_INSTS_VMEM_RD") and c.get("SQ_INST_LEVEL_VMEM"):
        c["avg_vmem_latency_quadcycles"] = round(c["SQ_INST_LEVEL_VMEM"] / max(c.get("SQ_INSTS_VMEM", c["SQ_INSTS_VMEM_RD"]), 1), 1)
    if c.get("SQ_INSTS_SMEM") and c.get("SQ_INST_LEVEL_SMEM"):
        c["avg_smem_latency_quadcycles"] = round(c["SQ_INST_LEVEL_SMEM"] / c["SQ_INSTS_SMEM"], 1)
    if c.get("TCP_TCC_READ_REQ_sum") and c.get("TCP_TCC_READ_REQ_LATENCY_sum"):
        c["avg_l1_to_l2_read_latency_cycles"] = round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], 1)
    if dur.get(k) and c.get("SQ_ACTIVE_INST_VALU"):
        ns = sum(dur[k]) / len(dur[k])
        c["kernel_us_in_sq2_pass"] = round(ns / 1e3, 1)
        c["valu_busy_frac"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * ns * 2.4), 3)
    rep["kernels"][k] = c
json.dump(rep, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(rep)[:6000])
