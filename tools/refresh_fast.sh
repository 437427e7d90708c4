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
# rocprofv3's own per-kernel averages over the FROZEN training scenes: the durations bench.py's roofline.{avg_us, frac} are built from
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/scene_stats" -o t -- python "$R/tools/scene_times.py" "$R/profiles/scenes/live_s1_a.npz" \
    "$R/profiles/scenes/live_s1_b.npz" > "$O/scene_stats.log" 2>&1)
# counters of the raster kernels on the FROZEN scenes (deterministic; the passes above see the live step's lottery scene)
for sc in live_s1_a survey_8d; do
  if [ "$sc" = survey_8d ]; then T="tools/kernels.py 3"; else T="tools/scene_times.py profiles/scenes/$sc.npz --iters 3"; fi
  PMC_TARGET="$T" PMC_GROUPS=sq1,sq2,tcc1 python tools/pmc_passes.py "$O/pmc_$sc" > "$O/pmc_$sc.log" 2>&1
done
python bench.py > "$O/bench_full.json" 2> "$O/bench_full.err"
python bench.py --model 0 --cpu-baseline 0 --fixed-scene 0 > "$O/bench_hotpath_only.json" 2> "$O/bench_hot.err"
python - "$O" <<'PY'
# steady state from the per-dispatch trace: the LAST 8 of the 13 steps (the first steps hold MIOpen's solver search -- seconds of
# naive reference convolutions -- and the allocator warm-up), steps delimited by the textured forward's dispatches
import csv, glob, json, sys, collections
out = sys.argv[1]
rows = []
for fn in glob.glob(out + "/stats/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_raster_forward<1" in r["Kernel_Name"]]
n = 8
sel = rows[marks[-n - 1]:marks[-1]] if len(marks) > n else rows
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
with open(out + "/steady_kernel_stats.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "CallsPerStep", "AverageUs", "UsPerStep"])
    for k, (ns, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        w.writerow([k, "%.2f" % (c / n), "%.2f" % (ns / c / 1e3), "%.2f" % (ns / n / 1e3)])
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / n / 1e3 if sel else 0
json.dump({"steps": n, "kernel_us_per_step": sum(v[0] for v in agg.values()) / n / 1e3, "wall_us_per_step": span, "launches_per_step": len(sel) / n},
          open(out + "/steady_totals.json", "w"))
print(open(out + "/steady_totals.json").read())
PY
find "$O" -name "*counter_collection.csv" -delete; find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*.csv" -size +3M -delete
tail -2 "$O/bench_full.json" | cut -c1-1500
