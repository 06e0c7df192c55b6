#!/bin/bash
mkdir -p gpurun_out/r04
T2V_POOL_DGRAD_DB=1 timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "pool_conv or up_conv" > gpurun_out/r04/test31.log 2>&1
tail -1 gpurun_out/r04/test31.log
F="--steps 30 --warmup 5 --no_cpu_baseline --no_extra --no_hbm"
for v in 0 1 0 1; do
  T2V_POOL_DGRAD_DB=$v timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench31_$v.log 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r04/bench31_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$v', d['ms_per_step'], d['d_fwdbwd_roofline']['all_in']['wall_ms'], d['d_fwdbwd_roofline']['conv_kernels']['ms'], r['by_tile'].get('pool dgrad 64x64'))
PY
done
