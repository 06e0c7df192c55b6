#!/bin/bash
mkdir -p gpurun_out/r04
F="--steps 30 --warmup 5 --no_cpu_baseline --no_extra --no_hbm"
for v in 0 16 0 16; do
  T2V_STRIP3_DB=$v timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench28_$v.log 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r04/bench28_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$v', d['ms_per_step'], d['d_fwdbwd_roofline']['all_in']['wall_ms'], d['d_fwdbwd_roofline']['conv_kernels']['ms'], r['by_tile'].get('strip3 64x64'))
PY
done
T2V_STRIP3_DB=16 timeout -k 10 900 python -m pytest tests/test_models_gpu.py -x -q -m gpu > gpurun_out/r04/test28.log 2>&1
tail -2 gpurun_out/r04/test28.log
