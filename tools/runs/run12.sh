#!/bin/bash
# GPU box: the viscosity work list -- tests, apply / iteration times, time steps, rocprof of the timed loop
python -m pytest tests/test_viscosity_march_gpu.py tests/test_pressure_gpu.py tests/test_bench_size_oracle_gpu.py tests/test_viscosity_gpu.py tests/test_viscosity_resident_gpu.py tests/test_viscosity_fused_gpu.py tests/test_viscosity_rdx_gpu.py tests/test_viscosity_slab_gpu.py tests/test_history_envelope.py tests/test_timestep_gpu.py -x -q > gpurun_out/r3_t12.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t12.log; tail -8 gpurun_out/r3_t12.log
python tools/vapply_time.py 256 f32 list > gpurun_out/r3_vapply_list.log 2>&1
MFS_VISC_SPARSE=0 python tools/vapply_time.py 256 f32 nolist >> gpurun_out/r3_vapply_list.log 2>&1
python tools/vapply_time.py 256 f64 list >> gpurun_out/r3_vapply_list.log 2>&1
python tools/vapply_time.py 128 f32 list >> gpurun_out/r3_vapply_list.log 2>&1
grep -a tag gpurun_out/r3_vapply_list.log
MFS_PRECISION=fp32 python tools/bench_timestep.py 256 2 > gpurun_out/r3_ts256e.log 2>&1; tail -1 gpurun_out/r3_ts256e.log | cut -c1-600
python tools/bench_timestep.py 256 2 > gpurun_out/r3_ts256f.log 2>&1; tail -1 gpurun_out/r3_ts256f.log | cut -c1-600
python tools/bench_timestep.py 128 3 > gpurun_out/r3_ts128c.log 2>&1; tail -1 gpurun_out/r3_ts128c.log | cut -c1-500
bash tools/prof_bench.sh r03_prof --no-side-legs > /dev/null 2>&1; cat gpurun_out/r03_prof_loop_summary.txt | cut -c1-140 | head -12
