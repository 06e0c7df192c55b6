#!/bin/bash
mkdir -p gpurun_out/r04
for v in early late early late; do
  export T2V_STRIP3_DB=16 T2V_LIB=tools/libt2v_$v.so
  echo "DB16 $v"; timeout -k 10 200 python tools/ablate_strip3.py 2>&1 | grep flags
done
