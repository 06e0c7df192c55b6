#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 700 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/r04/test34.log 2>&1
tail -3 gpurun_out/r04/test34.log
F="--steps 30 --warmup 5 --no_cpu_baseline --no_extra --no_hbm"
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export T2V_NO_CONV_PW=1; else unset T2V_NO_CONV_PW; fi
  timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench34_$v.log 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r04/bench34_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('nopw=$v', d['ms_per_step'], d['d_fwdbwd_roofline']['all_in']['wall_ms'], r['gpu_ms_per_step'], r['launches_per_step'], {k:(v['launches_per_step'], round(v['gpu_ms_per_step'],3)) for k,v in r['by_tile'].items() if 'igemm' in k or 'pw' in k})
PY
done
