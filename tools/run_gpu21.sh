#!/bin/bash
set -x
mkdir -p gpurun_out/r04
F="--steps 30 --warmup 5 --no_cpu_baseline --no_extra --no_hbm"
for v in 8 4 8 4 6; do
  T2V_POOL_WGRAD_MINCPS=$v timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench21_$v.log 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r04/bench21_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['ms_per_step'], d['d_fwdbwd_roofline']['all_in']['wall_ms'], d['roofline']['wgrad'])
PY
done
