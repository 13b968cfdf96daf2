#!/bin/bash
bash tools/prof_cmd.sh cfg4 50 tools/cfg4_leg.py > gpurun_out/r3_cfg4_prof.txt 2>&1; cat gpurun_out/r3_cfg4_prof.txt | cut -c1-170; tail -2 gpurun_out/cfg4.log | cut -c1-400
MFS_SPARSE=0 python tools/cfg4_leg.py 2>&1 | tail -1 | cut -c1-300
