#!/usr/bin/env bash
# GPU box, ~5 GPU-minutes: the bench lines of the OTHER workloads and the fixed-scene kernel timings (tools/refresh_fast.sh has the
# default line, the kernel trace and the PMC passes).  Every line carries config.lib_build_id; tools/make_summary.py refuses lines of
# another build than the tree's.  Outputs: gpurun_out/refresh_slow/
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
O="$R/gpurun_out/refresh_slow"; rm -rf "$O"; mkdir -p "$O"
B="--cpu-baseline 0 --fixed-scene 0 --hot-path-sub 0"
python bench.py --steps 20 --warmup 5 --graph 0 $B > "$O/bench_full_eager.json" 2> "$O/bench_full_eager.err"                 # the eager step: host-enqueue bound
python bench.py --workload s2 --steps 10 --warmup 3 $B > "$O/bench_s2.json" 2> "$O/bench_s2.err"                              # BASELINE configs[2] per-GPU shape
python bench.py --workload s2 --image-size 512 --subdivide 4 --steps 5 --warmup 2 $B > "$O/bench_s2_cfg4.json" 2> "$O/bench_s2_cfg4.err"   # configs[3] shape
python bench.py --force-ddp 1 --steps 10 --warmup 5 $B > "$O/bench_ddp1.json" 2> "$O/bench_ddp1.err"                          # 1-rank RCCL: the all-reduce path
python bench.py --share-mask-render 0 --steps 20 --warmup 5 $B > "$O/bench_full_two_renders.json" 2> "$O/bench_two.err"       # A/B of DESIGN 4.5
python bench.py --workload s2 --share-mask-render 0 --steps 10 --warmup 3 $B > "$O/bench_s2_two_renders.json" 2> "$O/bench_s2_two.err"
: > "$O/kernels.jsonl"
for sc in "0.6 0.9" "0.95 1.05"; do UMR_SCALE="$sc" python tools/kernels.py 20 >> "$O/kernels.jsonl" 2>> "$O/kernels.err"; done
UMR_N=128 python tools/kernels.py 5 >> "$O/kernels.jsonl" 2>> "$O/kernels.err"        # train_s2's launch size
python tools/cold_cache.py > "$O/cold_cache.json" 2>> "$O/kernels.err"
python tools/concurrency.py > "$O/concurrency.json" 2>> "$O/kernels.err"
UMR_CFG4=1 python tools/sweep_fm.py kernel_only > "$O/kernel_only.log" 2>&1
for f in "$O"/bench_*.json; do python - "$f" <<'PY'
import json, sys, os
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %8.1f img/s %7.2f ms/step graph=%s" % (os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], d["config"]["hip_graph"]))
except Exception as e:
    print(os.path.basename(sys.argv[1]), "unreadable:", e)
PY
done
