#!/bin/bash
# GPU box: every round-3 evidence file of profiles/ from the tree as it is (outputs under gpurun_out/, copied by hand afterwards)
bash tools/prof_bench.sh r03_prof > /dev/null 2>&1
bash tools/pmc_bench.sh r03_pmc > /dev/null 2>&1
bash tools/pmc_bench.sh r03_pmc_dense --dense-coefficients --unfused > /dev/null 2>&1
bash tools/prof_visc.sh r03pv 256 > gpurun_out/r03_visc_kernel_stats_256.txt 2>&1
bash tools/pmc_visc.sh r03pmcv 256 > gpurun_out/r03_visc_pmc_256.txt 2>&1
bash tools/pmc_visc.sh r03pmcvd 256 MFS_VISC_COMPRESS=0 MFS_VISC_SPARSE=0 > gpurun_out/r03_visc_pmc_256_dense.txt 2>&1
export MFS_PRECISION=fp32
bash tools/prof_total.sh r03ts tools/bench_timestep.py 256 2 > gpurun_out/r03_ts256_prof.txt 2>&1
unset MFS_PRECISION
python tools/bench_timestep.py 256 2 > gpurun_out/r3_ts256_f64_final.log 2>&1
python tools/bench_timestep.py 128 3 > gpurun_out/r3_ts128_final.log 2>&1
python tools/run_notebook_scene.py 100 > gpurun_out/r3_nbscene_final.log 2>&1
python tools/particle_bench.py 256 5 > gpurun_out/r3_particles_final.log 2>&1
python tools/visc_small_iter.py 48 80 48 f64 > gpurun_out/r3_vsmall_final.log 2>&1
python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err
python bench.py --steps 20 --warmup 5 --no-side-legs --no-cpu-baseline > gpurun_out/r3_bench_final_short.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench_final.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["parity_check"]["ok"])
for k,v in d["viscosity"].items(): print(k, v.get("us_per_iteration"), v.get("error"))
print(d["config4_rank_share"].get("us_per_iteration"), d["timestep_128"].get("s_per_step"))
PY
cat gpurun_out/r03_prof_loop_summary.txt | cut -c1-140 | head -8
