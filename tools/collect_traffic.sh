#!/usr/bin/env bash
# HBM traffic of the raster kernels inside the bench workload (GPU box).  PMC passes are separate runs with
# --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2; no sys/hip tracing with --pmc).
# Usage: tools/collect_traffic.sh <outdir> [bench args...]
set -euo pipefail
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$1"; shift
mkdir -p "$OUT"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -o t -- python "$R/bench.py" --steps 5 --warmup 2 --cpu-baseline 0 "$@" > "$OUT/$c.log" 2>&1
done
# calibration: a kernel with a KNOWN byte count and the raster kernels' per-lane access width (dword loads)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/CALIB" -o t -- python "$R/tools/prof_traffic.py" > "$OUT/CALIB.log" 2>&1
python "$R/tools/traffic_report.py" "$OUT" "$@"
