#!/usr/bin/env bash
# GPU box, round 4 first call: instruction-cost additions, baseline kernel timings, and the counters VERDICT r3 asked for.
set -uo pipefail
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$R/gpurun_out/r4_diag"
mkdir -p "$OUT"
cd "$R"
( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_ubench2.hip -o /tmp/valu_ubench2 && timeout 300 /tmp/valu_ubench2 ) > "$OUT/ubench2.log" 2>&1
timeout 600 python tools/sweep_fm.py r4_baseline > "$OUT/sweep_baseline.json" 2> "$OUT/sweep_baseline.err"
timeout 200 python tools/r4/step_kernels.py 10 0.6 0.9 > "$OUT/step_kernels.json" 2>&1
timeout 200 python tools/r4/step_kernels.py 10 0.95 1.05 >> "$OUT/step_kernels.json" 2>&1
UMR_DEBUG_SET=exact_edges=0 timeout 200 python tools/r4/step_kernels.py 10 0.6 0.9 >> "$OUT/step_kernels.json" 2>&1
timeout 900 python tools/r4/pmc_passes.py "$OUT/pmc" 3 0.6 0.9 > "$OUT/pmc.log" 2>&1
tail -c 3000 "$OUT/ubench2.log"; cat "$OUT/step_kernels.json"; tail -c 1500 "$OUT/pmc.log"
