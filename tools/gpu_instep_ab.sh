#!/usr/bin/env bash
# GPU box: the round-4 tree (gpurun_scratch/r4: its library AND its Python) and this tree on the SAME box, same command, eager:
# per-kernel HIP-event averages of the raster launches inside the training step (bench.py's roofline pass) and step times.
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/r5_instep"; mkdir -p "$O"
for rep in 1 2; do
  (cd "$R/gpurun_scratch/r4" && python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --fixed-scene 0 > "$O/r4_$rep.json" 2> "$O/r4_$rep.err")
  (cd "$R" && python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --fixed-scene 0 --graph 0 --hot-path-sub 0 > "$O/r5_$rep.json" 2> "$O/r5_$rep.err")
done
(cd "$R/gpurun_scratch/r4" && python bench.py --model 0 --graph 1 --cpu-baseline 0 --fixed-scene 0 > "$O/r4_hot.json" 2> "$O/r4_hot.err")
(cd "$R" && python bench.py --model 0 --cpu-baseline 0 --fixed-scene 0 > "$O/r5_hot.json" 2> "$O/r5_hot.err")
python - "$O" <<'PY'
import json, sys, glob, os
for fn in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(fn).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(fn), "unreadable", e); continue
    r = d["roofline"]
    ks = [r.get("avg_us"), r["forward_kernel"].get("avg_us"), r["silhouette_forward"].get("avg_us"), r["silhouette_backward"].get("avg_us")]
    print("%-14s %7.1f img/s %6.2f ms/step | bwd %s fwd %s sil_fwd %s sil_bwd %s | sum %s" % (
        os.path.basename(fn), d["value"], d["ms_per_step"], *["%.1f" % k if k else "-" for k in ks], "%.1f" % sum(k for k in ks if k)))
PY
