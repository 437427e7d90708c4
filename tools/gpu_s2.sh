#!/usr/bin/env bash
# GPU box: train_s2 (BASELINE configs[2] per-GPU shape) and the configs[3] raster shape from one HIP graph; one line each
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
O="$R/gpurun_out/r5_s2"; mkdir -p "$O"
timeout 900 python bench.py --workload s2 --steps 10 --warmup 3 --cpu-baseline 0 --fixed-scene 0 > "$O/bench_s2.json" 2> "$O/bench_s2.err"
timeout 900 python bench.py --workload s2 --image-size 512 --subdivide 4 --steps 5 --warmup 2 --cpu-baseline 0 --fixed-scene 0 > "$O/bench_s2_cfg4.json" 2> "$O/bench_s2_cfg4.err"
python - "$O" <<'PY'
import json, sys
for n in ("bench_s2", "bench_s2_cfg4"):
    try:
        d = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        c = d["config"]; r = d["roofline"]
        print(n, round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms graph", c["hip_graph"], "eager_host", c.get("eager_host_enqueue_ms_per_step"),
              "bwd us", r.get("avg_us"), "frac", r.get("frac"), "fwd us", r["forward_kernel"].get("avg_us"), "raster us/step", r.get("raster_kernels_us_per_step"), "launches", r.get("raster_launches_per_step"))
    except Exception as e:
        print(n, "unreadable", e); print(open(sys.argv[1] + "/" + n + ".err").read()[-1500:])
PY
