#!/usr/bin/env bash
# GPU box: freeze what the training step really renders.  Two default bench.py runs (two trajectories: float-atomic summation order
# alone makes them diverge) each dump the shared render's and the unseen-view silhouette's inputs at the first profile step;
# tools/scene_times.py then times the raster launches on each capture with the captured texels / gradient and with seeded noise.
# Outputs: gpurun_out/scenes/live_s1_{a,b}.npz (+ bench lines), gpurun_out/scenes/scene_times.jsonl
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
O="$R/gpurun_out/scenes"; mkdir -p "$O"
for t in a b; do
  python bench.py --cpu-baseline 0 --hot-path-sub 0 --fixed-scene 0 --capture-scene "$O/live_s1_$t.npz" > "$O/bench_$t.json" 2> "$O/bench_$t.err"
  tail -c 600 "$O/bench_$t.json"; echo
done
python tools/scene_times.py --real "$O/live_s1_a.npz" "$O/live_s1_b.npz" | tee "$O/scene_times.jsonl"
python tools/kernels.py 20 | tee -a "$O/scene_times.jsonl"
ls -la "$O"
