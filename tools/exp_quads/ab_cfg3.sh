#!/usr/bin/env bash
# GPU box: the step kernels at BASELINE configs[3]'s raster shape (N = 16, IS 1024, 5120 faces), product vs experimental libraries
set -uo pipefail
R="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$R"
OUT="$R/gpurun_out/r4_ab"; mkdir -p "$OUT"; LOG="$OUT/quads_cfg3.jsonl"; : > "$LOG"
export UMR_AG=1
timeout 300 python tools/r4/step_kernels.py 6 0.6 0.9 16 1024 4 2>/dev/null | grep '^{' >> "$LOG"
for lib in "$@"; do UMR_LIB_FILE="$lib" timeout 300 python tools/r4/step_kernels.py 6 0.6 0.9 16 1024 4 2>/dev/null | grep '^{' >> "$LOG"; done
cat "$LOG"
