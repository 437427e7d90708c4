#!/usr/bin/env bash
# HBM traffic of the raster kernels inside the bench workload (GPU box).  PMC passes are separate runs with
# --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2; no sys/hip tracing with --pmc).
# Usage: tools/collect_traffic.sh <outdir> [bench args...]
set -euo pipefail
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$(realpath -m "$1")"; shift
mkdir -p "$OUT"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -o t -- python "$R/bench.py" --steps 5 --warmup 2 --cpu-baseline 0 --fixed-scene 0 --graph 0 --hot-path-sub 0 "$@" > "$OUT/$c.log" 2>&1
done
# VALU roofline of the same kernels: two SQ passes (8 slots each), again separate runs with --kernel-trace only
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/SQ1" -o t -- python "$R/bench.py" --steps 5 --warmup 2 --cpu-baseline 0 --fixed-scene 0 --graph 0 --hot-path-sub 0 "$@" > "$OUT/SQ1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
  --output-format csv -d "$OUT/SQ2" -o t -- python "$R/bench.py" --steps 5 --warmup 2 --cpu-baseline 0 --fixed-scene 0 --graph 0 --hot-path-sub 0 "$@" > "$OUT/SQ2.log" 2>&1
# calibration: a kernel with a KNOWN byte count and the raster kernels' per-lane access width (dword loads)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/CALIB" -o t -- python "$R/tools/prof_traffic.py" > "$OUT/CALIB.log" 2>&1
python "$R/tools/traffic_report.py" "$OUT" "$@"
