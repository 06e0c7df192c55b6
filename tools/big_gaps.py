"""List the large idle gaps of a kernel trace window with the kernels on both sides (developer tool)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:50]) for r in csv.DictReader(open(f)))
t0, t1 = rows[0][0], rows[-1][1]
lo = t0 + (t1 - t0) * float(sys.argv[2])
hi = lo + float(sys.argv[3]) * 1e6
win = [r for r in rows if r[0] >= lo and r[1] <= hi]
last = win[0]
n = 0
for r in win[1:]:
    g = r[0] - last[1]
    if g > 200e3:
        print('%8.1f us gap at +%.2f ms   after %-40s before %s   (%d kernels since previous big gap)' % (g / 1e3, (r[0] - lo) / 1e6, last[2], r[2], n))
        n = 0
    n += 1
    if r[1] > last[1]:
        last = r
