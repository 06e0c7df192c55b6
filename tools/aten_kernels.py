"""Which ATen (non-t2v) kernels run inside the replayed steps: full template names + counts from a kernel-trace CSV."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'at::' in n or 'rocclr' in n or 'Cijk' in n:
        agg[n[:400]][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        agg[n[:400]][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
    print('%8.3f ms  x%6d  %s\n' % (v[0] / 1e6, v[1], k))
