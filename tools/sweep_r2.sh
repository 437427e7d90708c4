#!/usr/bin/env bash
# GPU box: A/B of the per-mesh super-block bins (round 2).  Output: gpurun_out/r2i/sweep.log
set -uo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"; O=gpurun_out/r2i; mkdir -p $O
(UMR_SB=1 timeout 200 python tools/sweep_fm.py "bins" 2>/dev/null | tail -1; UMR_SB=0 timeout 200 python tools/sweep_fm.py "nobins" 2>/dev/null | tail -1) > $O/sweep.log 2>&1
cat $O/sweep.log
