#!/bin/bash
mkdir -p gpurun_out/r04
F="--steps 20 --warmup 5 --no_cpu_baseline --no_extra --no_hbm --no_d_roofline"
for v in 1 0; do
  if [ $v = 1 ]; then export T2V_NO_CONV_PW=1; else unset T2V_NO_CONV_PW; fi
  T2V_PROF_DUMP=gpurun_out/r04/launches35_$v.csv timeout -k 10 400 python bench.py $F > gpurun_out/r04/bench35_$v.log 2>&1
  python tools/launch_table.py gpurun_out/r04/launches35_$v.csv 5 > gpurun_out/r04/shapes35_$v.txt 2>&1
done
python - <<'PY'
def load(f):
    d={}
    for l in open(f):
        if l.startswith('fwd/dgrad'):
            p=l.split()
            if p[6] != '1' and p[7] != '1': pass
            key=tuple(p[3:9])
            if p[6]=='1':   # taps == 1
                us=float(l.split('|')[1].split()[2]); n=float(l.split('|')[1].split()[0])
                d[key]=(p[1],n,us)
    return d
a=load('gpurun_out/r04/shapes35_1.txt'); b=load('gpurun_out/r04/shapes35_0.txt')
t0=t1=0
for k in sorted(a, key=lambda k:-a[k][1]*a[k][2]):
    if k in b:
        t0+=a[k][1]*a[k][2]; t1+=b[k][1]*b[k][2]
        print('%-34s n=%.0f %-6s %6.1f -> %-6s %6.1f us'%(' '.join(k), a[k][1], a[k][0], a[k][2], b[k][0], b[k][2]))
print(t0,t1)
PY
