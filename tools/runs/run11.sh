#!/bin/bash
# GPU box: tests of the sparse lists, the full bench line, rocprof stats, PMC passes, particle bench
python -m pytest tests/test_pressure_gpu.py tests/test_timestep_gpu.py tests/test_p2p_gpu.py tests/test_bench_rehearsal_gpu.py tests/test_resident_gpu.py tests/test_density_gpu.py -x -q > gpurun_out/r3_t11.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t11.log; tail -6 gpurun_out/r3_t11.log
python bench.py > gpurun_out/r3_bench4.json 2> gpurun_out/r3_bench4.err; tail -c 600 gpurun_out/r3_bench4.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench4.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["effective_all_cells"]["frac"], d["parity_check"]["history_max_rel_dev"])
print(d.get("sparse_lists"))
print(d["f64_state"]["ms_per_step"], d["roofline"]["dense"]["plain_stencil_apply"]["frac"], d.get("cpu_baseline"))
PY
bash tools/prof_bench.sh r03_prof --no-side-legs > /dev/null 2>&1
bash tools/pmc_bench.sh r03_pmc > /dev/null 2>&1
bash tools/pmc_bench.sh r03_pmc_dense --dense-coefficients --unfused > /dev/null 2>&1
python tools/particle_bench.py 256 5 > gpurun_out/r3_particles5.log 2>&1; tail -1 gpurun_out/r3_particles5.log | cut -c1-700
