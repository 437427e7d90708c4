#!/bin/bash
# GPU box: the open item of DESIGN.md section 5 -- bench.py at the configs[3] shape ended ~8 % of its processes on a non-finite loss in
# round 3's first builds.  N processes on the tripwire build (-DUMR_TRAP=1: every kernel of the library reports the first
# non-finite value it reads or writes, tools/r3/bench_trap.py); one line per process: discarded runs, earliest site.
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/nan; mkdir -p $O
N=${1:-30}
for i in $(seq 1 $N); do
  timeout 300 python tools/r3/bench_trap.py --workload s2 --image-size 512 --subdivide 4 --steps 30 --warmup 2 --cpu-baseline 0 > $O/run_$i.json 2> $O/run_$i.err
  rc=$?
  d=$(grep -o '"discarded_nonfinite_runs": [0-9]*' $O/run_$i.json | head -1)
  s=$(grep -h "bench_trap: earliest" $O/run_$i.err | tail -1)
  echo "run $i rc $rc $d | $s" | tee -a $O/summary.log
  if [ $rc -eq 0 ] && ! grep -q "site [1-9]" $O/run_$i.err && echo "$d" | grep -q ": 0"; then rm -f $O/run_$i.json $O/run_$i.err; fi
done
