#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$R"
O="$R/gpurun_out/adopt"; mkdir -p "$O"
tools/collect_traffic.sh "$O/traffic" > "$O/traffic.log" 2>&1; echo "traffic rc=$?"
find "$O/traffic" -name "*.csv" -delete
cp "$O/traffic/traffic.json" "$O/traffic.json"
python - "$O/traffic.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(j["build_id"], j["raster_backward_bytes_per_launch"])
for k, e in j["kernels"].items():
    print(k[:90], e["launches"], round(e["hbm_bytes_per_launch"] / 1e6, 1), round(((e.get("valu") or {}).get("issued_wave_instr") or 0) / 1e6, 1))
PY
