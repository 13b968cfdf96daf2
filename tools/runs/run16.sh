#!/bin/bash
bash tools/prof_visc.sh r03pv 256 > gpurun_out/r03_visc_kernel_stats_256b.txt 2>&1; cat gpurun_out/r03_visc_kernel_stats_256b.txt | cut -c1-150
python - <<'PY'
import os, sys, json
sys.path[:0] = ["python-fluid-simulation_amd", "."]
import torch, bench
out = bench.viscosity_leg(torch, torch.device("cuda:0"), 256, 60, False, "fp32", 0.0)
print(json.dumps({k: out[k] for k in ("us_per_iteration", "sparse_lists", "class_census")}))
PY
