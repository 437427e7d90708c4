#!/usr/bin/env bash
# GPU box, round 5 call 1: targeted parity tests of the new kernels, fixed-scene kernel timings (product / debug keys / experimental
# builds), bench lines (graph default, eager, train_s2 from a graph)
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$R"
O="$R/gpurun_out/r5_run1"; mkdir -p "$O"
timeout 1200 python -m pytest tests/test_gpu_round5.py \
  "tests/test_gpu_round4.py::test_alpha_geometry_render_routes_gradients_like_the_two_renders" \
  "tests/test_gpu_round4.py::test_shared_mask_render_step_equals_the_two_render_step" \
  tests/test_gpu_parity.py -x -q > "$O/tests.log" 2>&1; echo "tests rc=$?" >> "$O/tests.log"
tail -5 "$O/tests.log"
K="$O/kernels.jsonl"; : > "$K"
for sc in "0.6 0.9" "0.95 1.05"; do UMR_SCALE="$sc" timeout 300 python tools/r5/kernels.py 20 >> "$K" 2>> "$O/kernels.err"; done
for s in face_order=0 face_order_group=8 face_order_group=4; do UMR_DEBUG_SET=$s timeout 300 python tools/r5/kernels.py 20 >> "$K" 2>> "$O/kernels.err"; done
for lib in umr_amd/lib/exp/libumr_hip_*.so; do UMR_LIB_FILE=$lib timeout 300 python tools/r5/kernels.py 20 >> "$K" 2>> "$O/kernels.err"; done
timeout 300 python tools/r5/kernels.py 20 >> "$K" 2>> "$O/kernels.err"
cat "$K"
timeout 900 python bench.py --steps 20 --warmup 5 > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --graph 0 --cpu-baseline 0 --hot-path-sub 0 --fixed-scene 0 > "$O/bench_eager.json" 2> "$O/bench_eager.err"
timeout 900 python bench.py --workload s2 --steps 10 --warmup 3 --cpu-baseline 0 --fixed-scene 0 > "$O/bench_s2.json" 2> "$O/bench_s2.err"; echo "s2 rc=$?"
tail -3 "$O/bench_s2.err"
python - "$O" <<'PY'
import json, sys
for n in ("bench_default", "bench_eager", "bench_s2"):
    try:
        d = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        c = d["config"]
        print(n, round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms graph", c["hip_graph"], "host", round(c["host_enqueue_ms_per_step"], 2),
              "eager_host", c.get("eager_host_enqueue_ms_per_step"), "hot", c.get("hot_path_images_per_s"), c.get("hot_path_error"),
              "roofline us", d["roofline"].get("avg_us"), "frac", d["roofline"].get("frac"), "raster us/step", d["roofline"].get("raster_kernels_us_per_step"))
    except Exception as e:
        print(n, "unreadable", e)
PY
