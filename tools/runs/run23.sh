#!/bin/bash
bash tools/ab_libs.sh "product nomask product nomask product nomask" bench.py --no-cpu-baseline --no-side-legs --no-f64-line --steps 600 | cut -c1-260
export MFS_PRECISION=fp32
bash tools/ab_libs.sh "product nomask" tools/bench_timestep.py 256 2 | cut -c1-420
