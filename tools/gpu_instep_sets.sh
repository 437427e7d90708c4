#!/usr/bin/env bash
# GPU box: in-step effect of umr_debug_set switches: the default bench line's roofline pass (eager) and the hot-path graph, per
# switch list of AB_SETS ("a=1;b=2"), alternating with the default.   usage: AB_SETS="face_order_group=8" gpu_instep_sets.sh [repeats]
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
N="${1:-2}"
IFS=';' read -ra SETS <<< "${AB_SETS:-}"
show() { python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
ks = [r.get('avg_us'), r['forward_kernel'].get('avg_us'), r['silhouette_forward'].get('avg_us'), r['silhouette_backward'].get('avg_us')]
print('%-26s %-6s %7.1f img/s %6.2f ms | bwd %s fwd %s sil_fwd %s sil_bwd %s' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], *['%.1f' % k if k else '-' for k in ks]))" "$1" "$2"; }
for i in $(seq 1 "$N"); do
  for s in "" "${SETS[@]}"; do
    UMR_DEBUG_SET="$s" python bench.py --model 0 --cpu-baseline 0 --fixed-scene 0 2>/dev/null | show "[$s]" hot
    UMR_DEBUG_SET="$s" python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --fixed-scene 0 --hot-path-sub 0 2>/dev/null | show "[$s]" full
  done
done
