#!/bin/bash
python -m pytest tests/test_viscosity_march_gpu.py tests/test_viscosity_slab_gpu.py tests/test_viscosity_gpu.py tests/test_bench_size_oracle_gpu.py tests/test_history_envelope.py tests/test_pressure_gpu.py -x -q > gpurun_out/r3_t26.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t26.log; tail -3 gpurun_out/r3_t26.log
python tools/vapply_time.py 256 f32 qmask 2>&1 | tail -1 | cut -c1-200
python tools/vapply_time.py 256 f64 qmask 2>&1 | tail -1 | cut -c1-200
MFS_PRECISION=fp32 python tools/bench_timestep.py 256 2 2>&1 | tail -1 | cut -c1-420
