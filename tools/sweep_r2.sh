#!/usr/bin/env bash
# GPU box: A/B of two-waves-per-face in the face-major backward (round 2).  Output: gpurun_out/r2j/sweep.log
set -uo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"; O=gpurun_out/r2j; mkdir -p $O
(UMR_SPLIT=1 timeout 200 python tools/sweep_fm.py "split" 2>/dev/null | tail -1; UMR_SPLIT=2 timeout 200 python tools/sweep_fm.py "split" 2>/dev/null | tail -1) > $O/sweep.log 2>&1
cat $O/sweep.log
