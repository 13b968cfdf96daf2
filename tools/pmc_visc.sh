#!/bin/bash
# GPU box: SQ / TCC counters of the viscosity per-iteration kernels (separate --pmc passes).
# usage: tools/pmc_visc.sh <tag> <N> [env assignments, e.g. MFS_VISC_TILED=0]
set -e
TAG=${1:-pmcv}; N=${2:-256}; shift 2 || true
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
cd /tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"
P2="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"
P3="FETCH_SIZE"      # gfx950: reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): doubled below
P4="WRITE_SIZE"
i=0
for C in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_p$i -- python3 $R/tools/bench_viscosity.py $N ${PMC_DT:-f32} 20 > $R/gpurun_out/${TAG}_p$i.log 2>&1 || true
done
python3 - $R/gpurun_out/${TAG} <<'PY'
import csv, glob, sys, collections
base = sys.argv[1]
for i in (1, 2, 3, 4):
    fs = glob.glob(f"{base}_p{i}/**/*counter_collection.csv", recursive=True)
    if not fs: print("no counters for pass", i); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "vcg_apply" in name or "update_xr" in name: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        avg = {c: round(sum(v) / len(v)) for c, v in d.items()}
        if "FETCH_SIZE" in avg: avg["read_bytes(2 x FETCH_SIZE KiB)"] = 2 * avg["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in avg: avg["write_bytes"] = avg["WRITE_SIZE"] * 1024
        print(k[:70], avg, "launches", len(next(iter(d.values()))))
PY
rm -rf $R/gpurun_out/${TAG}_p1 $R/gpurun_out/${TAG}_p2 $R/gpurun_out/${TAG}_p3 $R/gpurun_out/${TAG}_p4
