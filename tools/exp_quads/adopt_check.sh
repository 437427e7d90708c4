#!/usr/bin/env bash
# GPU box, after the kernels of tools/exp_quads went into the product: differential test against the previous build of the library,
# the GPU tests of the shared render, smoke, one bench line.  The previous build: git worktree add /tmp/prev ffbff5a && python
# tools/exp_quads/build_gpu.py /tmp/prev/umr_amd/csrc r4frozen  (-> umr_amd/lib/exp/libumr_hip_r4frozen.so, build id cd11e46b...)
set -uo pipefail
R="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$R"
O="$R/gpurun_out/adopt"; mkdir -p "$O"
timeout 400 python tools/exp_quads/differential_gpu.py umr_amd/lib/exp/libumr_hip_r4frozen.so > "$O/differential.log" 2>&1; echo "differential rc=$?"; tail -3 "$O/differential.log"
timeout 400 python -m pytest tests/test_gpu_round4.py -x -q -k "alpha_geometry or shared_mask" 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --steps 10 --warmup 6 --cpu-baseline 0 > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
python - "$O/bench.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
r = j["roofline"]
print(round(j["value"], 1), round(j["ms_per_step"], 3), r["kernel"], round(r["avg_us"], 1), r["alg_bytes_per_launch"], round(r["frac"], 4), "sil bwd", round(r["silhouette_backward"]["avg_us"], 1), r["silhouette_backward"]["launches"], "traffic", r["traffic"], r["fixed_scene_us"])
PY
