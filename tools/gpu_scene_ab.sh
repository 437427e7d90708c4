#!/usr/bin/env bash
# GPU box: raster launch timings on the FROZEN scenes (profiles/scenes/*.npz = what the training step really renders, + the SURVEY 8d
# scene; tools/scene_times.py) of the product library, of every experimental build under umr_amd/lib/exp/ and of the product library
# with each umr_debug_set key list of AB_SETS ("a=1,b=2;c=3"); product first and last.  usage: tools/gpu_scene_ab.sh <tag> [iters]
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
TAG="$1"; IT="${2:-20}"
O="$R/gpurun_out/scene_ab"; mkdir -p "$O"; K="$O/$TAG.jsonl"; : > "$K"
run() { timeout 600 python tools/scene_times.py --iters "$IT" >> "$K" 2>> "$O/$TAG.err"; }
run
IFS=';' read -ra SETS <<< "${AB_SETS:-}"
for s in "${SETS[@]}"; do [ -n "$s" ] && UMR_DEBUG_SET="$s" run; done
for lib in umr_amd/lib/exp/libumr_hip_*.so; do [ -e "$lib" ] && UMR_LIB_FILE="$lib" run; done
run
python - "$K" <<'PY'
import json, sys, re
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
short = lambda k: {"textured_forward_p2f_vis_pool": "fwd", "texel_gradient_backward": "bwd_tex", "silhouette_forward": "sil_fwd", "silhouette_backward": "sil_bwd",
                   "shared_render_backward_one_pass": "AGP", "shared_render_forward_packed_state": "fwd_pk",
                   "shared_render_backward_one_pass_planar_state": "AG_planar", "vertex_gradient_backward": "bwd_vert"}.get(re.sub(r"_N\d+$", "", k), k[:9])
print("%-26s %-30s %-10s %s" % ("lib", "set", "scene", "us per launch"))
for r in rows:
    print("%-26s %-30s %-10s " % (r["lib"][:26], r["set"][:30], r["scene"]) + " ".join("%s %.1f" % (short(k), v) for k, v in r["us_per_launch"].items()))
PY
