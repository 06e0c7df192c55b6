#!/bin/bash
set -x
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split_k_in_one_launch or conv_fwd_bwd or pool_conv_group or bf16_cases or conv_grouped" > gpurun_out/r04/test16.log 2>&1
tail -5 gpurun_out/r04/test16.log
F="--steps 30 --warmup 5 --no_cpu_baseline --no_extra --no_hbm"
T2V_NO_FUSED_SPLITK=1 timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench16_legacy.log 2>&1
tail -c 300 gpurun_out/r04/bench16_legacy.log; echo
timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench16_fused.log 2>&1
tail -c 300 gpurun_out/r04/bench16_fused.log; echo
python - <<'PY'
import json
for f in ('legacy','fused'):
    for l in open('gpurun_out/r04/bench16_%s.log'%f):
        if l.startswith('{'):
            d=json.loads(l); print(f, d['ms_per_step'], d['d_fwdbwd_roofline']['all_in'], d['roofline']['splitk_reduce'])
PY
