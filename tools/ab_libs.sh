#!/bin/bash
# GPU box: the same measurement with several builds of the library on ONE box (box-to-box spread is ~10 %).
# usage: tools/ab_libs.sh "<lib names under mfs/variants, or 'product'>" <script> [args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
LIBS=$1; shift
for L in $LIBS; do
  if [ "$L" = product ]; then unset MFS_LIB; else export MFS_LIB=$R/python-fluid-simulation_amd/mfs/variants/libmfs_hip_$L.so; fi
  echo "== $L"
  timeout -k 10 300 python3 "$R/$1" "${@:2}" 2>/dev/null < /dev/null | tail -n 1 | cut -c1-600
done
