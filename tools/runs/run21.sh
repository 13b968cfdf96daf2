#!/bin/bash
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r3_bench_rep$i.json 2> gpurun_out/r3_bench_rep$i.err
python - $i <<'PY'
import json, sys
d=json.loads(open("gpurun_out/r3_bench_rep%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], "cfg4", d["config4_rank_share"].get("us_per_iteration"), "visc", d["viscosity"]["n256"].get("us_per_iteration"), "ts", d["timestep_128"].get("s_per_step"), "jac", d["jacobi_preconditioned"].get("jacobi",{}).get("us_per_iteration"))
PY
done
