#!/usr/bin/env bash
# GPU box: step-kernel timings (tools/r4/step_kernels.py) for the product library and for each experimental library given.
# usage: tools/r4/ab.sh <tag> [lib.so ...]     env AB_SETS="a=1,b=2;c=3" adds runs of the product library with umr_debug_set keys
set -uo pipefail
R="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$R"
TAG="$1"; shift
OUT="$R/gpurun_out/r4_ab"; mkdir -p "$OUT"
LOG="$OUT/$TAG.jsonl"; : > "$LOG"
run() { for sc in "0.6 0.9" "0.95 1.05"; do timeout 300 python tools/r4/step_kernels.py 20 $sc 2>/dev/null | grep '^{' >> "$LOG"; done; }
run
IFS=';' read -ra SETS <<< "${AB_SETS:-}"
for s in "${SETS[@]}"; do [ -n "$s" ] && UMR_DEBUG_SET="$s" run; done
for lib in "$@"; do UMR_LIB_FILE="$lib" run; done
run
cat "$LOG"
