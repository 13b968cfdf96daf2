#!/bin/bash
# container: an A/B build of libmfs_hip.so with extra compiler flags, next to the product library (never loaded unless
# MFS_LIB points at it).   usage: tools/build_variant.sh NAME "-DMFS_VM_CELL_GROUP=0 ..."
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2
D=$R/python-fluid-simulation_amd/csrc
B=$D/_build_$NAME
mkdir -p $B $R/python-fluid-simulation_amd/mfs/variants
for f in $D/*.hip; do
  o=$B/$(basename $f .hip).o
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-array-bounds $FLAGS -c $f -o $o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/python-fluid-simulation_amd/mfs/variants/libmfs_hip_$NAME.so $B/*.o
rm -rf $B
echo built python-fluid-simulation_amd/mfs/variants/libmfs_hip_$NAME.so
