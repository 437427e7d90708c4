#!/usr/bin/env bash
# GPU box: fixed-scene kernel timings (tools/kernels.py) of the product library, of every experimental build under
# umr_amd/lib/exp/ and of the product library with each umr_debug_set key list of AB_SETS ("a=1,b=2;c=3"); product first and last.
# usage: tools/gpu_ab.sh <tag> [iters]     env AB_SCALES="0.6 0.9;0.95 1.05" (default: the first only)
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
TAG="$1"; IT="${2:-20}"
O="$R/gpurun_out/r5_ab"; mkdir -p "$O"; K="$O/$TAG.jsonl"; : > "$K"
IFS=';' read -ra SCALES <<< "${AB_SCALES:-0.6 0.9}"
run() { for sc in "${SCALES[@]}"; do UMR_SCALE="$sc" timeout 300 python tools/kernels.py "$IT" >> "$K" 2>> "$O/$TAG.err"; done; }
run
IFS=';' read -ra SETS <<< "${AB_SETS:-}"
for s in "${SETS[@]}"; do [ -n "$s" ] && UMR_DEBUG_SET="$s" run; done
for lib in umr_amd/lib/exp/libumr_hip_*.so; do [ -e "$lib" ] && UMR_LIB_FILE="$lib" run; done
run
python - "$K" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
keys = list(rows[0]["us_per_launch"]) if rows else []
import re
short = lambda k: {"textured_forward_p2f_vis_pool": "fwd", "texel_gradient_backward": "bwd_tex", "silhouette_forward": "sil_fwd", "silhouette_backward": "sil_bwd",
                   "shared_render_backward_one_pass": "AGP", "shared_render_forward_packed_state": "fwd_pk",
                   "shared_render_backward_one_pass_planar_state": "AG_planar"}.get(re.sub(r"_N\d+$", "", k), k[:9])
print("%-28s %-22s %-10s " % ("lib", "set", "scale") + " ".join("%9s" % short(k) for k in keys))
for r in rows:
    print("%-28s %-22s %-10s " % (r["lib"], r["set"], r["scale"]) + " ".join("%9.1f" % r["us_per_launch"][k] for k in keys))
PY
