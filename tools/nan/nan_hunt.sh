#!/bin/bash
# GPU box: N bench.py processes at the BASELINE configs[3] shape (train_s2, 512^2 images = 1024^2 render, 5120 faces, bs 16) on the
# tripwire build of the CURRENT library; one line per process: discarded runs, earliest non-finite site, non-finite terms.
# usage: nan_hunt.sh [N=20] [steps=60]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/nan; rm -rf $O; mkdir -p $O
N=${1:-20}; STEPS=${2:-60}
# OLD=1: the tripwire build with round 3's EARLY settings -- tile cull band without the reference-noise widening, no thin-face
# route, nearest edge by true distance everywhere (exact_edges off): the configuration the non-finite runs were seen on
if [ "${OLD:-0}" = "1" ]; then
  export TRAP_LIB=trap_old UMR_DEBUG_SET=exact_edges=0
  python tools/build_variant.py trap_old -DUMR_TRAP=1 -DTILE_CULL_NOISE=0.f -DTHIN_FACE_H=0.f > $O/build.log 2>&1 || { echo "trap build failed"; tail -5 $O/build.log; exit 1; }
  echo "== OLD settings (no cull widening, no thin-face route, exact_edges 0)" | tee -a $O/summary.log
else
  python tools/build_variant.py trap -DUMR_TRAP=1 > $O/build.log 2>&1 || { echo "trap build failed"; tail -5 $O/build.log; exit 1; }
fi
for i in $(seq 1 $N); do
  timeout 600 python tools/nan/bench_trap.py --workload s2 --image-size 512 --subdivide 4 --steps $STEPS --warmup 2 --profile-steps 1 --cpu-baseline 0 > $O/run_$i.json 2> $O/run_$i.err
  rc=$?
  d=$(grep -o '"discarded_nonfinite_runs": [0-9]*' $O/run_$i.json | head -1)
  s=$(grep -h "bench_trap: earliest\|bench_trap: first raster" $O/run_$i.err | tr '\n' ' ')
  echo "run $i rc $rc $d | $s" | tee -a $O/summary.log
  if [ $rc -eq 0 ] && ! grep -q "site [1-9]" $O/run_$i.err && echo "$d" | grep -q ": 0"; then rm -f $O/run_$i.json $O/run_$i.err; fi
done
