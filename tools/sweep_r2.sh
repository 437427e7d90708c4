#!/usr/bin/env bash
# GPU box: A/B of compile-time variants of the raster kernels (round 2).  Output: gpurun_out/r2e/sweep.log
set -uo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"; O=gpurun_out/r2e; mkdir -p $O
for v in "-DFM_VREC=0" "-DFM_VREC=1"; do
  python -c "import sys; from umr_amd.build import build; build(force=True, verbose=False, extra_flags=sys.argv[1].split())" "$v" || continue
  timeout 200 python tools/sweep_fm.py "$v" 2>/dev/null | tail -1
done > $O/sweep.log 2>&1
python -c "from umr_amd.build import build; build(force=True, verbose=False)"
cat $O/sweep.log
