#!/bin/bash
# GPU box: rocprofv3 kernel stats of tools/bench_viscosity.py.  usage: tools/prof_visc.sh <tag> <N> [ENV=val ...]
set -e
TAG=${1:-pv}; N=${2:-256}; shift 2 || true
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/tools/bench_viscosity.py $N f32 100 > $R/gpurun_out/$TAG.log 2>&1
F=$(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "mfs::" not in n: continue
    n = n.split("(")[0].replace("void ", "")
    if int(r["Calls"]) < 50: continue
    print(f"{n[:58]:58s} {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}  max {float(r['MaxNs'])/1e3:9.2f}")
PY
rm -rf $R/gpurun_out/$TAG
