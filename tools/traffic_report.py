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
bw = [v for k, v in rep["kernels"].items() if "k_raster_backward_fm<1" in k]
rep["raster_backward_bytes_per_launch"] = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in bw) / max(1, sum(v["launches"] for v in bw)) if bw else None
rep["workload"] = [16, 256, 3, model]
json.dump(rep, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(rep, indent=1)[:3000])
