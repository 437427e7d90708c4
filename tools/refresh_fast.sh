#!/usr/bin/env bash
# GPU box, <= 4 GPU-minutes: the part of profiles/ that a KERNEL CHANGE invalidates, to be re-run after the last kernel commit --
#   1. rocprofv3 --kernel-trace --stats of the bench step (eager launches: every kernel a dispatch of its own), 13 steps
#   2. the PMC passes of the step's raster launches (FETCH_SIZE / WRITE_SIZE / two SQ groups + calibration) -> traffic.json,
#      stamped with the library's build id (bench.py attaches it only to that build)
#   3. the default bench line (whole step from one HIP graph) carrying those figures, and the hot-path-only line
# Outputs: gpurun_out/refresh_fast/; `python tools/make_summary.py rNN --fast` then copies the summaries into profiles/ and
# writes profiles/rNN_SUMMARY.md -- refusing anything measured on another build than the tree's.
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/refresh_fast"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "from umr_amd import _lib; print(_lib.build_id())" > "$O/build_id.txt"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o t -- python "$R/bench.py" --steps 10 --warmup 3 --profile-steps 0 --graph 0 \
    --cpu-baseline 0 --hot-path-sub 0 --fixed-scene 0 > "$O/stats.log" 2>&1)
tools/collect_traffic.sh "$O/traffic" > "$O/traffic.log" 2>&1
cp "$O/traffic/traffic.json" profiles/traffic.json
python bench.py > "$O/bench_full.json" 2> "$O/bench_full.err"
python bench.py --model 0 --cpu-baseline 0 --fixed-scene 0 > "$O/bench_hotpath_only.json" 2> "$O/bench_hot.err"
python - "$O" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
rows = []
for fn in glob.glob(out + "/stats/*kernel_stats.csv"):
    rows += list(csv.DictReader(open(fn)))
ours = {r["Name"].replace("(anonymous namespace)::", ""): {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
        for r in rows if "k_raster" in r["Name"]}
json.dump(ours, open(out + "/raster_kernel_stats.json", "w"), indent=1)
print(json.dumps(ours, indent=1))
PY
find "$O" -name "*counter_collection.csv" -delete; find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*.csv" -size +3M -delete
tail -2 "$O/bench_full.json" | cut -c1-1500
