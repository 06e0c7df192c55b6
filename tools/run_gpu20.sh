#!/bin/bash
set -x
mkdir -p gpurun_out/r04
F="--steps 30 --warmup 5 --no_cpu_baseline --no_extra --no_hbm"
for v in 768 1024 1536 768 1024; do
  T2V_SPLIT_TARGET=$v timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench20_$v.log 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r04/bench20_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['ms_per_step'], d['d_fwdbwd_roofline']['all_in']['wall_ms'], d['roofline']['gpu_ms_per_step'], d['roofline']['splitk_reduce'])
PY
done
