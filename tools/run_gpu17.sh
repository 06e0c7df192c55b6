#!/bin/bash
set -x
mkdir -p gpurun_out/r04
T2V_PROF_DUMP=gpurun_out/r04/d_launches.csv timeout -k 10 300 python tools/d_roofline.py --iters 3 > gpurun_out/r04/d17.log 2>&1
tail -c 900 gpurun_out/r04/d17.log; echo
python tools/launch_table.py gpurun_out/r04/d_launches.csv 3 > gpurun_out/r04/d_launch_shapes.txt 2>&1
head -50 gpurun_out/r04/d_launch_shapes.txt
timeout -k 10 900 python tools/parity_steps.py 100 gpurun_out/r04/r04_parity_100_steps.json > gpurun_out/r04/parity17.log 2>&1
tail -5 gpurun_out/r04/parity17.log
