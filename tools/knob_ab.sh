#!/bin/bash
# GPU box: one env knob, several values, alternating processes of one tool.  usage: tools/knob_ab.sh KNOB "v1 v2 ..." <script> [args]
K=$1; VALS=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in $VALS; do
  echo "== $K=$v"; env $K=$v timeout -k 10 300 python3 "$R/$1" "${@:2}" 2>/dev/null < /dev/null | tail -n 1 | cut -c1-420
done; done
