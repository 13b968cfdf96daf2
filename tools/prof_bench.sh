#!/bin/bash
# GPU box: rocprofv3 kernel-trace + stats of the default bench command; summary -> gpurun_out/<tag>/
# usage: tools/prof_bench.sh <tag> [bench args...]
set -e
TAG=${1:-prof}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$TAG.log 2>&1
F=$(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$F" "$R/gpurun_out/$TAG.log" > $R/gpurun_out/${TAG}_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline", " ".join(sys.argv[3:]))
print("# bench line:", [l for l in open(sys.argv[2]) if l.startswith("{")][-1].strip())
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'pct':>6s}")
for r in rows:
    n = r["Name"]
    if "mfs::" not in n: continue
    n = n.split("(")[0].replace("void ", "")
    print(f"{n[:60]:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['Percentage']):6.2f}")
PY
cat $R/gpurun_out/${TAG}_summary.txt
# ... and of the timed loop alone (no roofline / side legs: the stencil kernel's average is then the timed loop's own)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_loop -- python3 $R/bench.py --no-cpu-baseline --timed-loop-only "$@" > $R/gpurun_out/${TAG}_loop.log 2>&1
F=$(find $R/gpurun_out/${TAG}_loop -name "*kernel_stats.csv" | head -1)
python3 - "$F" "$R/gpurun_out/${TAG}_loop.log" > $R/gpurun_out/${TAG}_loop_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --timed-loop-only")
print("# bench line:", [l for l in open(sys.argv[2]) if l.startswith("{")][-1].strip()[:700])
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'pct':>6s}")
for r in rows:
    n = r["Name"]
    if "mfs::" not in n: continue
    n = n.split("(")[0].replace("void ", "")
    print(f"{n[:60]:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['Percentage']):6.2f}")
PY
cat $R/gpurun_out/${TAG}_loop_summary.txt
rm -rf $R/gpurun_out/${TAG}_loop $R/gpurun_out/$TAG
