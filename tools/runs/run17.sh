#!/bin/bash
# GPU box: 128-unknown live chunks, viscosity slab lists -- tests, iteration times, time steps, bench
python -m pytest tests/test_pressure_gpu.py tests/test_viscosity_march_gpu.py tests/test_viscosity_slab_gpu.py tests/test_p2p_gpu.py tests/test_bench_size_oracle_gpu.py tests/test_timestep_gpu.py tests/test_history_envelope.py tests/test_density_gpu.py tests/test_viscosity_gpu.py -x -q > gpurun_out/r3_t17.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t15.log; tail -8 gpurun_out/r3_t15.log
python tools/vapply_time.py 256 f32 c32 > gpurun_out/r3_vapply_c32.log 2>&1
python tools/vapply_time.py 256 f64 c32 >> gpurun_out/r3_vapply_c32.log 2>&1
python tools/vapply_time.py 128 f32 c32 >> gpurun_out/r3_vapply_c32.log 2>&1
grep -a tag gpurun_out/r3_vapply_c32.log
MFS_PRECISION=fp32 python tools/bench_timestep.py 256 2 > gpurun_out/r3_ts256h.log 2>&1; tail -1 gpurun_out/r3_ts256g.log | cut -c1-600
python tools/bench_timestep.py 128 3 > gpurun_out/r3_ts128e.log 2>&1; tail -1 gpurun_out/r3_ts128d.log | cut -c1-500
python bench.py --no-cpu-baseline --no-side-legs > gpurun_out/r3_bench8.json 2> gpurun_out/r3_bench8.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench8.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["sparse_lists"])
PY
bash tools/prof_visc.sh r03pv 256 > gpurun_out/r03_visc_kernel_stats_256c.txt 2>&1; cat gpurun_out/r03_visc_kernel_stats_256c.txt | cut -c1-150
export MFS_PRECISION=fp32
bash tools/prof_total.sh r03ts tools/bench_timestep.py 256 2 > gpurun_out/r03_ts256_prof_b.txt 2>&1; cat gpurun_out/r03_ts256_prof_b.txt | cut -c1-170
