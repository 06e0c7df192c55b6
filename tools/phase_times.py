"""Where one graph-replayed iteration spends its GPU time, by PHASE (developer tool), from a rocprofv3 --kernel-trace CSV:
    python tools/phase_times.py <rocprof dir> <ms_per_step>
A step is cut at marker kernels: pyramid_gather_k (start of the iteration: the real pyramid), rsgan_mean_multi_k (a loss is
evaluated: the forward passes before it are done), rsgan_mean_multi_bwd_k (its backward starts), adam_multi_k (an optimiser runs).
Prints, for the steps inside the densest window, the mean time from each marker to the next and the five heaviest kernels of
each phase."""
import bisect
import collections
import csv
import glob
import sys

d = sys.argv[1]
ms_step = float(sys.argv[2])
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:70]) for r in csv.DictReader(open(f))]
rows.sort()
# densest window of 8 steps (the replays), as tools/gap_analysis.py
t0, t1 = rows[0][0], rows[-1][1]
wlen = 8 * ms_step * 1e6
starts = [r[0] for r in rows]
cum = [0]
for s_, e_, _ in rows:
    cum.append(cum[-1] + e_ - s_)
best, w0, cand = -1, t0, t0
while cand + wlen <= t1:
    i0, i1 = bisect.bisect_left(starts, cand), bisect.bisect_left(starts, cand + wlen)
    if cum[i1] - cum[i0] > best:
        best, w0 = cum[i1] - cum[i0], cand
    cand += ms_step * 1e6 / 4
win = [r for r in rows if r[0] >= w0 and r[1] <= w0 + wlen]
wstarts = [r[0] for r in win]
MARK = ('pyramid_gather_k', 'rsgan_mean_multi_k', 'rsgan_mean_multi_bwd_k', 'adam_multi_k')
# phases: consecutive markers (adam launches back to back count as one)
phases = collections.OrderedDict()
cur_name, cur_start, cur_k = None, None, None
seq = []
last_mark = None
for s_, e_, name in win:
    if name in MARK and name != last_mark:             # (markers launched several times in a row count once)
        seq.append((name, s_))
    if name in MARK:
        last_mark = name
    elif name not in ('adam_tick_k',):
        last_mark = None
# index of each step start
idx = [i for i, (n_, _) in enumerate(seq) if n_ == 'pyramid_gather_k']
labels = None
acc = collections.defaultdict(float)
heavy = collections.defaultdict(lambda: collections.defaultdict(float))
nsteps = 0
for a, b in zip(idx, idx[1:]):
    marks = seq[a:b + 1]
    names = ['%d:%s->%s' % (i, marks[i][0].replace('_k', ''), marks[i + 1][0].replace('_k', '')) for i in range(len(marks) - 1)]
    if labels is None:
        labels = names
    if names != labels:
        continue
    nsteps += 1
    for i in range(len(marks) - 1):
        lo, hi = marks[i][1], marks[i + 1][1]
        acc[names[i]] += (hi - lo) / 1e6
        for s_, e_, name in win[bisect.bisect_left(wstarts, lo):bisect.bisect_left(wstarts, hi)]:
            heavy[names[i]][name] += (e_ - s_) / 1e6
print('%d steps; phases (marker -> next marker), mean ms per step:' % nsteps)
tot = 0.0
for n_ in labels or []:
    t = acc[n_] / max(nsteps, 1)
    tot += t
    top = sorted(heavy[n_].items(), key=lambda kv: -kv[1])[:6]
    print('  %-58s %6.3f ms | %s' % (n_, t, ', '.join('%s %.2f' % (k.replace('void ', '')[:34], v / nsteps) for k, v in top)))
print('  sum %.3f ms' % tot)
