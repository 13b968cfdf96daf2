#!/bin/bash
# GPU box: round-3 profiles -- viscosity kernel stats / PMC (lists on, dense), time-step kernel totals, notebook scene, bench lines
bash tools/prof_visc.sh r03pv 256 > gpurun_out/r03_visc_kernel_stats_256.txt 2>&1
bash tools/pmc_visc.sh r03pmcv 256 > gpurun_out/r03_visc_pmc_256.txt 2>&1
bash tools/pmc_visc.sh r03pmcvd 256 MFS_VISC_COMPRESS=0 MFS_VISC_SPARSE=0 > gpurun_out/r03_visc_pmc_256_dense.txt 2>&1
export MFS_PRECISION=fp32
bash tools/prof_total.sh r03ts tools/bench_timestep.py 256 2 > gpurun_out/r03_ts256_prof.txt 2>&1
unset MFS_PRECISION
python tools/run_notebook_scene.py 100 > gpurun_out/r3_nbscene6.log 2>&1; tail -1 gpurun_out/r3_nbscene6.log | cut -c1-400
python bench.py --no-cpu-baseline > gpurun_out/r3_bench6.json 2> gpurun_out/r3_bench6.err; tail -c 300 gpurun_out/r3_bench6.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs > gpurun_out/r3_bench6_short.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench6.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k,v in d["viscosity"].items(): print(k, v.get("us_per_iteration"), v.get("sparse_lists"), v.get("error"))
d=json.loads(open("gpurun_out/r3_bench6_short.json").read().strip().splitlines()[-1])
print("short", d["value"], d["ms_per_step"], d["timed_blocks"])
PY
cat gpurun_out/r03_visc_kernel_stats_256.txt | cut -c1-150; cat gpurun_out/r03_visc_pmc_256.txt | cut -c1-300; cat gpurun_out/r03_ts256_prof.txt | cut -c1-200
