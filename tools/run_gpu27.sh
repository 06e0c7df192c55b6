#!/bin/bash
mkdir -p gpurun_out/r04
for v in 16 32; do
T2V_STRIP3_DB=$v timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv_fwd_bwd or conv_grouped or relu_conv or double_backward or even_frames" > gpurun_out/r04/test27_$v.log 2>&1
tail -1 gpurun_out/r04/test27_$v.log
done
for v in 0 16 32 0 16; do
  export T2V_STRIP3_DB=$v
  echo "DB=$v"; timeout -k 10 200 python tools/ablate_strip3.py 2>&1 | grep flags
done
