"""Top conv launch shapes by time from T2V_PROF_DUMP."""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in rows:
    k = (int(r['kind']), int(r['M']), int(r['Cin']), int(r['Cout']), int(r['taps']), int(r['groups']), int(r['S']))
    a = agg[k]
    a[0] += float(r['ms']); a[1] += float(r['flops']); a[2] += 1
tot = sum(a[0] for a in agg.values())
print('total %.2f ms/step' % (tot / steps))
print('kind       M   Cin  Cout taps grp   S |  n/step  ms/step   avg us  TFLOP/s')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print('%4d %7d %5d %5d %4d %3d %3d | %6.1f %8.3f %8.1f %8.1f' % (k + (a[2] / steps, a[0] / steps, 1e3 * a[0] / a[2], a[1] / a[0] / 1e9 if a[0] else 0)))
