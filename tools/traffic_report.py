"""Post-process tools/collect_traffic.sh: per-launch HBM bytes of the raster kernels, corrected as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes (FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE
reports 1/2 of the bytes of a coalesced stream -- verified here on a kernel with a known read volume and dword
loads; WRITE_SIZE is taken as is)."""
import collections, csv, json, os, sys

out = sys.argv[1]
args = sys.argv[2:]
model = not ("--model" in args and args[args.index("--model") + 1] == "0")


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


fetch = per_kernel(os.path.join(out, "FETCH_SIZE", "t_counter_collection.csv"), "FETCH_SIZE")
write = per_kernel(os.path.join(out, "WRITE_SIZE", "t_counter_collection.csv"), "WRITE_SIZE")
calib = per_kernel(os.path.join(out, "CALIB", "t_counter_collection.csv"), "FETCH_SIZE")
known = 2 * 128 * 512 * 512 * 4
cal_kb = [v for k, vs in calib.items() if "k_iou_partial" in k for v in vs]
factor = known / (1024.0 * sum(cal_kb) / len(cal_kb)) if cal_kb else 2.0
rep = {"calibration": {"kernel": "k_iou_partial", "known_read_bytes": known, "FETCH_SIZE_KB": sum(cal_kb) / max(1, len(cal_kb)),
                       "fetch_correction_factor": factor}, "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if "(anonymous namespace)::k_raster" not in k:
        continue
    f = fetch.get(k, []); w = write.get(k, [])
    fkb = sum(f) / max(1, len(f)); wkb = sum(w) / max(1, len(w))
    rep["kernels"][k] = {"launches": len(f), "FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb,
                         "hbm_bytes_per_launch": fkb * 1024 * factor + wkb * 1024}
# ---- VALU roofline (passes SQ1 / SQ2): per launch, issued wave-instructions, lane use, fraction of the fp32 vector peak
# (157.3 TFLOP/s = 1228.9 G wave64-instructions/s: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles, MI355X_MICROARCH.md)
VALU_PEAK = 157.3e12 / 128.0


def counters(pass_dir):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    path = os.path.join(out, pass_dir, "t_counter_collection.csv")
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def durations(pass_dir):
    d = collections.defaultdict(list)
    path = os.path.join(out, pass_dir, "t_kernel_trace.csv")
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d


sq1, sq2, dur = counters("SQ1"), counters("SQ2"), durations("SQ1")
mean = lambda v: sum(v) / len(v) if v else None
for k, e in rep["kernels"].items():
    a, b = sq1.get(k, {}), sq2.get(k, {})
    issued, thread_cyc, act_valu = mean(a.get("SQ_INSTS_VALU", [])), mean(b.get("SQ_THREAD_CYCLES_VALU", [])), mean(b.get("SQ_ACTIVE_INST_VALU", []))
    ns = mean(dur.get(k, []))
    if issued and ns:
        lane_use = thread_cyc / (act_valu * 64.0) if thread_cyc and act_valu else None
        wc = mean(a.get("SQ_WAVE_CYCLES", []))
        e["valu"] = {"issued_wave_instr": issued, "salu_wave_instr": mean(a.get("SQ_INSTS_SALU", [])),
                     "lane_use": lane_use, "useful_lane_instr": issued * 64.0 * lane_use if lane_use else None,
                     "kernel_us_in_pmc_pass": ns / 1e3, "wave_instr_per_s": issued / (ns * 1e-9),
                     "frac_of_peak": issued / (ns * 1e-9) / VALU_PEAK, "peak_wave_instr_per_s": VALU_PEAK,
                     "waves": mean(a.get("SQ_WAVES", [])),
                     # SQ_ACTIVE_INST_VALU counts quad-cycles (guide, PMC section): x 4 = SIMD cycles with a VALU instruction in the pipe
                     "busy_cycles_per_instr": (4.0 * act_valu / issued) if act_valu else None,
                     "busy_frac": (4.0 * act_valu / (1024.0 * ns * 1e-9 * 2.4e9)) if act_valu else None}
        wa, wi, ac = mean(b.get("SQ_WAIT_ANY", [])), mean(b.get("SQ_WAIT_INST_ANY", [])), mean(b.get("SQ_ACTIVE_INST_ANY", []))
        if wa is not None and wi is not None and ac is not None and (wa + wi + ac) > 0:
            tot = wa + wi + ac                                     # disjoint buckets of a wave's life (guide, PMC section)
            e["valu"].update(wave_wait_frac=wa / tot, wave_issue_stall_frac=wi / tot, wave_active_frac=ac / tot)
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from umr_amd import _lib
    rep["build_id"] = _lib.build_id()          # bench.py attaches these figures only to the library they were measured on
except Exception as ex:                        # noqa: BLE001
    rep["build_id"] = "unknown (%s)" % ex
bw = [v for k, v in rep["kernels"].items() if "k_raster_backward_fm_agp<1" in k] or \
     [v for k, v in rep["kernels"].items() if "k_raster_backward_fm_ag<1" in k] or \
     [v for k, v in rep["kernels"].items() if "k_raster_backward_fm<1" in k]      # the step's dominant backward launch
rep["raster_backward_bytes_per_launch"] = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in bw) / max(1, sum(v["launches"] for v in bw)) if bw else None
rep["workload"] = [16, 256, 3, model]
json.dump(rep, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(rep, indent=1)[:3000])
