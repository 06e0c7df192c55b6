#!/bin/bash
mkdir -p gpurun_out/r04
F="--steps 20 --warmup 5 --no_cpu_baseline --no_extra --no_hbm --no_d_roofline"
for v in 0 16; do
  T2V_STRIP3_DB=$v T2V_PROF_DUMP=gpurun_out/r04/launches33_$v.csv timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench33_$v.log 2>&1
  python tools/launch_table.py gpurun_out/r04/launches33_$v.csv 5 > gpurun_out/r04/shapes33_$v.txt 2>&1
done
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        if l.startswith('fwd/dgrad') and ' strip3 ' in l:
            p=l.split()
            key=tuple(p[3:9])   # M Cin Cout taps mem S
            us=float(l.split('|')[1].split()[2]); n=float(l.split('|')[1].split()[0])
            d[key]=(n,us)
    return d
a=load('gpurun_out/r04/shapes33_0.txt'); b=load('gpurun_out/r04/shapes33_16.txt')
tot0=tot1=0
for k in sorted(a, key=lambda k:-a[k][0]*a[k][1]):
    if k in b:
        n,u0=a[k]; u1=b[k][1]; tot0+=n*u0; tot1+=n*u1
        print('%-40s n=%.0f  %7.1f -> %7.1f us  %+5.1f%%'%(' '.join(k), n, u0, u1, 100*(u1-u0)/u0))
print('total', tot0, tot1)
PY
