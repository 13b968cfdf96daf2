#!/bin/bash
# GPU box: HBM-side counters of the viscosity apply kernel for a set of library variants (MFS_LIB), separate --pmc passes.
# usage: tools/pmc_vapply.sh <tag> <N> variant [variant ...]     (variant "base" = the product library)
set -e
TAG=$1; N=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for V in "$@"; do
  L=""; [ "$V" != base ] && L=$R/python-fluid-simulation_amd/mfs/variants/libmfs_hip_$V.so
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    D=$R/gpurun_out/${TAG}_${V}_$(echo $C | cut -d' ' -f1)
    MFS_LIB=$L rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $R/tools/vapply_time.py $N f32 $V > $D.log 2>&1 || true
    python3 - "$D" "$V" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "vcg_apply_march" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {c: round(sum(v) / len(v)) for c, v in acc.items()}
if "FETCH_SIZE" in out: out["read_MB(2xFETCH_SIZE KiB)"] = round(2 * out["FETCH_SIZE"] * 1024 / 1e6, 1)
if "WRITE_SIZE" in out: out["write_MB"] = round(out["WRITE_SIZE"] * 1024 / 1e6, 1)
print(sys.argv[2], out)
PY
    rm -rf $D
  done
done
