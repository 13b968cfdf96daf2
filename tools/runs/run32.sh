#!/bin/bash
bash tools/ab_libs.sh "product chunk64 chunk16 product chunk64 chunk16" tools/vapply_time.py 256 f32 x | cut -c1-200
export MFS_PRECISION=fp32
bash tools/ab_libs.sh "product chunk64 chunk16" tools/bench_timestep.py 256 2 | cut -c1-420
