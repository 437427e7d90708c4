#!/usr/bin/env bash
# GPU box: regenerate everything under profiles/ that is measured (copied from gpurun_out/refresh afterwards).
set -uo pipefail
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/refresh"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python bench.py --steps 10 --warmup 5 --cpu-baseline 0 > /dev/null 2>&1        # MIOpen first-use search, page-in
python bench.py > "$O/bench_full.json" 2> "$O/bench_full.err"
python bench.py --model 0 --cpu-baseline 0 > "$O/bench_hotpath_only.json" 2> "$O/bench_hot.err"
python bench.py --workload s2 --cpu-baseline 0 --steps 10 --warmup 3 > "$O/bench_s2.json" 2> "$O/bench_s2.err"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o t -- python "$R/bench.py" --steps 10 --warmup 5 --cpu-baseline 0 > "$O/stats.log" 2>&1)
tools/collect_traffic.sh "$O/traffic" > "$O/traffic.log" 2>&1
python tools/microbench.py > "$O/microbench.log" 2>&1
python tools/microbench.py --alpha >> "$O/microbench.log" 2>&1
python tools/sweep_fm.py kernel_only > "$O/kernel_only.log" 2>&1
tools/pmc_raster.sh "$O/pmc_ts36" 64 3 512 36 > "$O/pmc_ts36.log" 2>&1
tools/pmc_raster.sh "$O/pmc_ts1" 64 3 512 1 > "$O/pmc_ts1.log" 2>&1
find "$O" -name "*.csv" -size +3M -delete      # keep the merged-back payload small (per-dispatch traces)
ls -la "$O"
