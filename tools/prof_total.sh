#!/bin/bash
# GPU box: total GPU-busy time of a python tool from rocprofv3 kernel stats (sum over all kernels), beside its own output.
# usage: tools/prof_total.sh <tag> <script> [args ...]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/$TAG.log 2>&1 < /dev/null
F=$(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"all kernels: {calls} launches, {tot/1e6:.2f} ms busy")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"  {r['Name'].split('(')[0].replace('void ', '')[:70]:70s} {r['Calls']:>7s} calls {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
grep -h '"scene"\|s_per_step' $R/gpurun_out/$TAG.log | tail -1 | cut -c1-400
rm -rf $R/gpurun_out/$TAG
