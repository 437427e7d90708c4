#!/usr/bin/env bash
# GPU box: start order of the one-pass backward's waves (k_face_order on / off, meshes per group), frame-filling scene
set -uo pipefail
R="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$R"
O="$R/gpurun_out/adopt"; mkdir -p "$O"; : > "$O/order.jsonl"
for s in "" "face_order=0" "face_order_group=4" "face_order_group=8"; do
  UMR_DEBUG_SET="$s" UMR_AG=1 timeout 200 python tools/r4/step_kernels.py 20 0.95 1.05 2>/dev/null | grep '^{' >> "$O/order.jsonl"
done
python - "$O/order.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l); u = j["us_per_launch"]
    print("%-22s texel-only %.1f  sil N32 %.1f  full %.1f  one-pass %.1f (err %.1e %.1e)" % (j["set"] or "default", u["tex_bwd_texel_only"], u["sil_bwd"], u["tex_bwd_full"], u["ag_bwd"], u["ag_err_vertex"], u["ag_err_texel"]))
PY
