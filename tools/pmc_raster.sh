#!/usr/bin/env bash
# GPU box: two SQ PMC passes (<= 8 counters each, --kernel-trace only) over a few raster launches at one shape.
# Usage: tools/pmc_raster.sh <outdir> N subdiv IS TS
set -euo pipefail
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$(realpath -m "$1")"; shift
mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS \
  --output-format csv -d "$OUT/p1" -o t -- python "$R/tools/prof_raster.py" "$@" > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS \
  --output-format csv -d "$OUT/p2" -o t -- python "$R/tools/prof_raster.py" "$@" > "$OUT/p2.log" 2>&1
python "$R/tools/pmc_report.py" "$OUT" "$1"
