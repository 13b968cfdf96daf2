#!/bin/bash
python -m pytest tests/test_pressure_gpu.py tests/test_p2p_gpu.py tests/test_density_gpu.py tests/test_timestep_gpu.py -x -q > gpurun_out/r3_t24.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t24.log; tail -3 gpurun_out/r3_t24.log
bash tools/ab_libs.sh "product nomask product nomask" bench.py --no-cpu-baseline --no-side-legs --no-f64-line --steps 600 | cut -c1-200
export MFS_PRECISION=fp32
bash tools/ab_libs.sh "product nomask" tools/bench_timestep.py 256 2 | cut -c1-420
