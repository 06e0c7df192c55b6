#!/bin/bash
set -x
mkdir -p gpurun_out/r04
F="--cond --bf16 --steps 20 --warmup 3 --no_cpu_baseline --no_extra --no_hbm --no_d_roofline"
for v in new1 new new1 new; do
  L=""; [ $v = new1 ] && L="tools/libt2v_new1.so"
  T2V_LIB=$L timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench19_$v.log 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r04/bench19_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['ms_per_step'], d['roofline']['wgrad'])
PY
done
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -x -q -m gpu > gpurun_out/r04/test19.log 2>&1
tail -3 gpurun_out/r04/test19.log
