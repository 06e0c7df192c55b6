"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the heaviest (kernel, grid) groups."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0])
byk = collections.defaultdict(lambda: [0, 0])
for r in rows:
    name = r['Kernel_Name'].split('(')[0][:70]
    key = (name, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'])
    dt = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[key][0] += dt
    agg[key][1] += 1
    byk[name][0] += dt
    byk[name][1] += 1
tot = sum(v[0] for v in agg.values())
print('kernel launches: %d   total GPU kernel time: %.2f ms   (/%g iterations = %.2f ms, %.0f launches per iteration)' %
      (len(rows), tot / 1e6, steps, tot / 1e6 / steps, len(rows) / steps))
print('\n-- by kernel')
for k, v in sorted(byk.items(), key=lambda kv: -kv[1][0])[:28]:
    print('%6.2f%% %9.1f us avg x%6d  %s' % (100 * v[0] / tot, v[0] / v[1] / 1e3, v[1], k))
print('\n-- heaviest (kernel, workgroups x,y,z)')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print('%6.2f%% %9.1f us avg x%6d  %s grid=(%s,%s,%s)' % (100 * v[0] / tot, v[0] / v[1] / 1e3, v[1], k[0][:60], k[1], k[2], k[3]))
