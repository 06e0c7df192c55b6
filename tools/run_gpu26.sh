#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/poison_check.py > gpurun_out/r04/poison.txt 2>&1; tail -6 gpurun_out/r04/poison.txt
timeout -k 10 300 python tools/det_check.py > gpurun_out/r04/det.txt 2>&1; tail -6 gpurun_out/r04/det.txt
F="--steps 30 --warmup 5 --no_cpu_baseline --no_extra --no_hbm"
timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench26.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/r04/bench26.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['d_fwdbwd_roofline']['all_in']['wall_ms'], d['d_fwdbwd_roofline']['conv_kernels']['ms'])
PY
