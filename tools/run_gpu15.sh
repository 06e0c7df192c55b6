#!/bin/bash
# A/B: weight-gradient launches on a side stream
set -x
mkdir -p gpurun_out/r04
F="--steps 30 --warmup 5 --no_cpu_baseline --no_extra --no_hbm"
python bench.py $F > gpurun_out/r04/bench15_main.log 2>&1
tail -c 400 gpurun_out/r04/bench15_main.log; echo
T2V_WGRAD_SIDE=1 timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench15_side.log 2>&1
tail -c 400 gpurun_out/r04/bench15_side.log; echo
T2V_WGRAD_SIDE=1 timeout -k 10 600 python -m pytest tests/test_models_gpu.py -x -q -m gpu > gpurun_out/r04/test15_side.log 2>&1
tail -5 gpurun_out/r04/test15_side.log
