#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/accum_sites.py 8 --cond > gpurun_out/r04/accum_cond.txt 2>&1
tail -40 gpurun_out/r04/accum_cond.txt
timeout -k 10 300 python tools/aten_sites.py --cond > gpurun_out/r04/aten_cond.txt 2>&1
tail -50 gpurun_out/r04/aten_cond.txt
timeout -k 10 600 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "second_stream" > gpurun_out/r04/test22.log 2>&1
tail -3 gpurun_out/r04/test22.log
