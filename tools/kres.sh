#!/bin/bash
# container: compile one .hip of csrc/ to gfx950 assembly and print the register / scratch footprint of the kernels
# whose mangled name contains PATTERN.   usage: tools/kres.sh mfs_visc.hip k_vcg_apply_march [extra hipcc flags]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; PAT=$2; shift 2 || true
OUT=/tmp/$(basename "$SRC" .hip).s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function --cuda-device-only -S -o "$OUT" "$@" \
  "$R/python-fluid-simulation_amd/csrc/$SRC" 2>&1 | grep -E "error|warning: v" -A3 | head -30 || true
python3 - "$OUT" "$PAT" <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S*' + re.escape(sys.argv[2]) + r'\S*)', s):
    blk = s[max(0, m.start() - 3000):m.start() + 1500]
    g = lambda k: (re.search(k + r':\s+(\d+)', blk) or [None, None])[1]
    print(m.group(1)[8:48], 'vgpr', g('.vgpr_count'), 'sgpr', g('.sgpr_count'), 'spill', g('.vgpr_spill_count'),
          'scratch', g('.private_segment_fixed_size'), 'lds', g('.group_segment_fixed_size'))
print('s_trap count in file:', s.count('s_trap'))
PY
