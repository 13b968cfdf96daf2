#!/bin/bash
# GPU box: lane-level dead masking in the pressure march
python -m pytest tests/test_pressure_gpu.py tests/test_p2p_gpu.py tests/test_density_gpu.py tests/test_bench_size_oracle_gpu.py tests/test_timestep_gpu.py tests/test_history_envelope.py tests/test_jacobi_gpu.py tests/test_resident_gpu.py tests/test_fuzz_gpu.py tests/test_bench_rehearsal_gpu.py -x -q > gpurun_out/r3_t22.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t22.log; tail -6 gpurun_out/r3_t22.log
python bench.py --no-cpu-baseline --no-side-legs > gpurun_out/r3_bench9.json 2> gpurun_out/r3_bench9.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench9.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["parity_check"]["ok"], d["f64_state"]["ms_per_step"])
PY
MFS_PRECISION=fp32 python tools/bench_timestep.py 256 2 > gpurun_out/r3_ts256i.log 2>&1; tail -1 gpurun_out/r3_ts256i.log | cut -c1-600
python tools/bench_timestep.py 128 3 > gpurun_out/r3_ts128f.log 2>&1; tail -1 gpurun_out/r3_ts128f.log | cut -c1-500
python tools/cfg4_leg.py 2>&1 | tail -1 | cut -c100-330
