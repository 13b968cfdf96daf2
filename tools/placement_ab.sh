#!/bin/bash
# GPU box: time per iteration of bench.py's timed loop over several PROCESSES per layout of the five CG vectors
# (separate allocations / one allocation with vector k shifted by k * S bytes).  usage: tools/placement_ab.sh "S1 S2 ..." [runs]
R=${GRAFT_REPO_ROOT:-/root/repo}
RUNS=${2:-4}
for S in $1; do
  for i in $(seq $RUNS); do
    if [ "$S" = sep ]; then unset MFS_BENCH_STAGGER; else export MFS_BENCH_STAGGER=$S; fi
    v=$(timeout -k 10 200 python3 $R/bench.py --timed-loop-only --no-cpu-baseline --steps 400 --warmup 30 2>/dev/null < /dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "layout=$S run=$i ms_per_step=$v"
  done
done
