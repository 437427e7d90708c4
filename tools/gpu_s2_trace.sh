#!/usr/bin/env bash
# GPU box: steady-state per-kernel table of the train_s2 step (eager, last 4 of 9 steps from the per-dispatch trace) ->
# gpurun_out/r5_s2_trace/steady_kernel_stats.csv (+ totals.json); `cp` into profiles/ as rNN_s2_kernel_stats_steady.csv
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/r5_s2_trace"; rm -rf "$O"; mkdir -p "$O"
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$O/raw" -o t -- python "$R/bench.py" --workload s2 --graph 0 --steps 6 --warmup 3 --profile-steps 0 \
    --cpu-baseline 0 --fixed-scene 0 --hot-path-sub 0 > "$O/bench.json" 2> "$O/bench.err")
python - "$O" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
rows = []
for fn in glob.glob(out + "/raw/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_edt_rows" in r["Kernel_Name"]]      # one per step
n = 4
sel = rows[marks[-n - 1]:marks[-1]]
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
with open(out + "/steady_kernel_stats.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "CallsPerStep", "AverageUs", "UsPerStep"])
    for k, (ns, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        w.writerow([k, "%.2f" % (c / n), "%.2f" % (ns / c / 1e3), "%.2f" % (ns / n / 1e3)])
tot = sum(v[0] for v in agg.values()) / n / 1e3
ours = sum(v[0] for k, v in agg.items() if k.replace("void ", "").startswith(("k_", "umr_k"))) / n / 1e3
raster = sum(v[0] for k, v in agg.items() if "k_raster" in k) / n / 1e3
json.dump({"steps": n, "kernel_us_per_step": tot, "libumr_us_per_step": ours, "raster_main_us_per_step": raster, "launches_per_step": len(sel) / n,
           "build_id": json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])["config"]["lib_build_id"]}, open(out + "/totals.json", "w"))
print(open(out + "/totals.json").read())
PY
rm -rf "$O/raw"
