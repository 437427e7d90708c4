#!/usr/bin/env bash
# GPU box: where does the time of the raster kernels go?  Compile-time variants (built HERE beforehand with
#   python tools/build_variants.py novisit="-DFM_NO_VISIT=1" nocull="-DFM_NO_CULL=1 -DFM_SKIP_EMPTY=0" fwdnovisit="-DFWD_NO_VISIT=1"
# ) timed with the library's own HIP events: default, backward without sub-tile visits (per-face set-up + culling pass),
# backward without the culling pass as well (per-face set-up + reductions), forward without face visits (binning, tile
# filter, epilogue stores).  One JSON line per variant: us per launch [forward, backward].
set -uo pipefail
cd "$(dirname "$0")/../.."
O=gpurun_out/split; mkdir -p "$O"
export UMR_CFG4="${UMR_CFG4:-0}"
for v in "" novisit nocull fwdnovisit; do
  UMR_LIB_VARIANT=$v timeout 300 python tools/sweep_fm.py "default-$v" 2>&1 | tail -1
done | tee "$O/split.log"
