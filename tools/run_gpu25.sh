#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv_fwd_bwd or deferred or wgrad or conv_grouped" > gpurun_out/r04/test25.log 2>&1
tail -3 gpurun_out/r04/test25.log
for f in 0 128; do
  T2V_LIB=tools/libt2v_ablation.so T2V_DEBUG_FLAGS=$f timeout -k 10 120 python tools/ablate_thin.py 2>&1 | grep flags
done
