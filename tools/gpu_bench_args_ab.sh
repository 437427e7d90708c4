#!/usr/bin/env bash
# GPU box: A/B of extra bench.py arguments on the default line (alternating runs).  usage: gpu_cl.sh "<args A>" "<args B>" [repeats]
set -u
export TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
A="$1"; B="$2"; N="${3:-2}"
for i in $(seq 1 "$N"); do
  for X in "$A" "$B"; do
    python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --fixed-scene 0 --hot-path-sub 0 --profile-steps 0 $X 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[%s]' % sys.argv[1], round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms graph', d['config']['hip_graph'], 'loss', round(d['config']['final_loss'], 4))" "$X"
  done
done
