#!/bin/bash
# GPU box: where the waves of the viscosity apply kernel spend their time (SQ / LDS / TA counters, separate --pmc passes)
# usage: tools/pmc_sq.sh <tag> <N> <dtype> variant [variant ...]
set -e
TAG=$1; N=$2; DT=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for V in "$@"; do
  L=""; [ "$V" != base ] && L=$R/python-fluid-simulation_amd/mfs/variants/libmfs_hip_$V.so
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
    D=$R/gpurun_out/${TAG}_${V}_x
    MFS_LIB=$L rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $R/tools/vapply_time.py $N $DT $V > $D.log 2>&1 || true
    python3 - "$D" "$V" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "vcg_apply_march" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {c: round(sum(v) / len(v)) for c, v in acc.items()})
PY
    rm -rf $D $D.log
  done
done
