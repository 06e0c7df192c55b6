#!/bin/bash
set -x
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "pool_conv or up_conv" > gpurun_out/r04/test18.log 2>&1
tail -3 gpurun_out/r04/test18.log
for v in new1 new; do
  L=""; [ $v = new1 ] && L="tools/libt2v_new1.so"
  T2V_LIB=$L T2V_PROF_DUMP=gpurun_out/r04/d_launches_$v.csv timeout -k 10 300 python tools/d_roofline.py --iters 3 > gpurun_out/r04/d18_$v.log 2>&1
  python tools/launch_table.py gpurun_out/r04/d_launches_$v.csv 3 > gpurun_out/r04/d_launch_shapes_$v.txt 2>&1
  grep "pool_dgrad\|pool_wgrad" gpurun_out/r04/d_launch_shapes_$v.txt
done
F="--steps 30 --warmup 5 --no_cpu_baseline --no_extra --no_hbm"
for v in new1 new new1 new; do
  L=""; [ $v = new1 ] && L="tools/libt2v_new1.so"
  T2V_LIB=$L timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench18_$v.log 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r04/bench18_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['ms_per_step'], d['d_fwdbwd_roofline']['all_in']['wall_ms'], d['d_fwdbwd_roofline']['conv_kernels']['ms'], d['roofline']['wgrad']['gpu_ms_per_step'])
PY
done
