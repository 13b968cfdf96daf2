#!/bin/bash
# GPU box: rocprofv3 kernel stats of an arbitrary python tool.  usage: tools/prof_cmd.sh <tag> <min calls> <script> [args ...]
# (ENV for the run: export before calling.)  Prints one line per mfs kernel; never reads stdin.
set -e
TAG=$1; MINC=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/$TAG.log 2>&1 < /dev/null
F=$(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] || { echo "no kernel_stats.csv under $TAG"; exit 1; }
python3 - "$F" "$MINC" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "mfs::" not in n: continue
    if int(r["Calls"]) < int(sys.argv[2]): continue
    short = n.split("(")[0].replace("void ", "")
    print(f"{short[:100]:100s} {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}  max {float(r['MaxNs'])/1e3:9.2f}")
PY
rm -rf $R/gpurun_out/$TAG
