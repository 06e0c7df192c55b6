#!/bin/bash
mkdir -p gpurun_out/r04
for f in 0 64 128 192 256 512 704 960; do
  T2V_LIB=tools/libt2v_ablation.so T2V_DEBUG_FLAGS=$f timeout -k 10 120 python tools/ablate_thin.py 2>&1 | grep flags
done
