#!/bin/bash
python -m pytest tests/test_viscosity_march_gpu.py tests/test_viscosity_slab_gpu.py tests/test_viscosity_gpu.py tests/test_bench_size_oracle_gpu.py tests/test_history_envelope.py tests/test_viscosity_resident_gpu.py tests/test_viscosity_fused_gpu.py tests/test_timestep_gpu.py -x -q > gpurun_out/r3_t27.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t27.log; tail -3 gpurun_out/r3_t27.log



