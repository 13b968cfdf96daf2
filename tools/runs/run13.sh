#!/bin/bash
# GPU box: slab sparse lists + level set -- tests, particle bench, full bench line
python -m pytest tests/test_p2p_gpu.py tests/test_bench_rehearsal_gpu.py tests/test_particles_gpu.py tests/test_pressure_gpu.py tests/test_timestep_gpu.py tests/test_density_gpu.py tests/test_jacobi_gpu.py tests/test_failure_gpu.py -x -q > gpurun_out/r3_t13.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t13.log; tail -8 gpurun_out/r3_t13.log
python tools/particle_bench.py 256 5 > gpurun_out/r3_particles6.log 2>&1; tail -1 gpurun_out/r3_particles6.log | cut -c1-400
python bench.py > gpurun_out/r3_bench5.json 2> gpurun_out/r3_bench5.err; tail -c 300 gpurun_out/r3_bench5.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench5.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
print("cfg4", {k: d["config4_rank_share"].get(k) for k in ("ms_per_step", "host_enqueue_ms_per_step", "parity_check", "error")})
v=d["viscosity"]
for k in v: print(k, v[k].get("us_per_iteration", v[k].get("ms_per_step")), v[k].get("error"))
print("ts", d["timestep_128"])
PY
