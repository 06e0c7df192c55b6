#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "cond or gen or attention2d or up_block" > gpurun_out/r04/test23.log 2>&1
tail -5 gpurun_out/r04/test23.log
timeout -k 10 300 python tools/accum_sites.py 8 --cond > gpurun_out/r04/accum_cond2.txt 2>&1
tail -12 gpurun_out/r04/accum_cond2.txt
timeout -k 10 300 python tools/aten_sites.py --cond > gpurun_out/r04/aten_cond2.txt 2>&1
tail -25 gpurun_out/r04/aten_cond2.txt
F="--cond --bf16 --steps 20 --warmup 3 --no_cpu_baseline --no_extra --no_hbm --no_d_roofline --no_roofline"
timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench23.log 2>&1
tail -c 600 gpurun_out/r04/bench23.log
