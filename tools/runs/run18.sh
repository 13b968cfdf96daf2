#!/bin/bash
python -m pytest tests/test_resident_gpu.py tests/test_viscosity_resident_gpu.py tests/test_edge_cases_gpu.py tests/test_failure_gpu.py tests/test_pressure_gpu.py tests/test_viscosity_gpu.py tests/test_density_gpu.py tests/test_notebook_gpu.py -x -q > gpurun_out/r3_t18.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t18.log; tail -5 gpurun_out/r3_t18.log
python tools/run_notebook_scene.py 100 > gpurun_out/r3_nbscene7.log 2>&1; tail -1 gpurun_out/r3_nbscene7.log | cut -c1-420
python tools/run_notebook_scene.py 300 > gpurun_out/r3_nbscene8.log 2>&1; tail -1 gpurun_out/r3_nbscene8.log | cut -c1-420
