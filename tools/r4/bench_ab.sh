#!/usr/bin/env bash
# GPU box: bench.py lines for a list of argument sets, one compact summary line each.  usage: bench_ab.sh "<args 1>" "<args 2>" ...
set -uo pipefail
R="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$R"
OUT="$R/gpurun_out/r4_bench"; mkdir -p "$OUT"
i=0
for a in "$@"; do
  i=$((i+1))
  python bench.py --cpu-baseline 0 $a > "$OUT/line_$i.json" 2> "$OUT/line_$i.err"
  python - "$OUT/line_$i.json" "$a" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); r=d["roofline"]
    f=lambda x: "%.1f/%s"%(x["avg_us"] or 0, x["launches"])
    print("%-60s %8.1f img/s %7.3f ms | tex bwd %s fwd %s sil fwd %s bwd %s | disc %s"%(sys.argv[2][:60], d["value"], d["ms_per_step"], f(r), f(r["forward_kernel"]), f(r["silhouette_forward"]), f(r["silhouette_backward"]), d["config"]["discarded_nonfinite_runs"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
