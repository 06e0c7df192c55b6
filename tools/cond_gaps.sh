#!/bin/bash
# developer tool: the idle gaps (> 10 us) of the text-conditioned iteration (bench.py --cond --bf16) under the profiler
R=$(pwd); O=$R/gpurun_out/r04; mkdir -p $O; rm -rf $O/trace_c
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_c -- python3 $R/bench.py --cond --bf16 --steps 10 --warmup 3 --no_cpu_baseline --no_roofline --no_d_roofline --no_extra --no_hbm > $O/trace_c.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r04/trace_c/*/*_kernel_trace.csv')[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:44]) for r in csv.DictReader(open(f)))
# last 60 ms of the trace = replayed steps
t1 = rows[-1][1]
win = [r for r in rows if r[0] >= t1 - 60e6]
last = win[0]; c = collections.Counter(); tot = collections.Counter()
for r in win[1:]:
    g = r[0] - last[1]
    if g > 10e3:
        c[(last[2], r[2])] += 1; tot[(last[2], r[2])] += g / 1e3
    if r[1] > last[1]: last = r
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print('%8.1f us in %3d gaps   after %-44s before %s' % (v, c[k], k[0], k[1]))
print('total gap us', sum(tot.values()), 'over 60 ms')
PY
rm -rf $O/trace_c
