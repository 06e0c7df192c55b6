"""Timeline of the graph-replayed steps from a rocprofv3 --kernel-trace CSV: GPU-busy time, idle gaps, and which
kernels the gaps follow. Usage: python tools/gap_analysis.py <rocprof dir> <ms_per_step> [n_steps_to_analyse]

[auto | start fraction]. The bench's eager warm-ups run before the timed region and its instrumented pass after it, so the
window of n steps is placed on the densest stretch of the trace (the graph replays) unless a start fraction is given."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
ms_step = float(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60]) for r in csv.DictReader(open(f))]
rows.sort()
# the replayed region is the densest stretch of the trace: with argv[4] = 'auto' (default) slide a window of n steps over the
# span and keep the start with the most kernel time inside; a number instead places the window at that fraction of the span
import bisect
t0, t1 = rows[0][0], rows[-1][1]
wlen = n * ms_step * 1e6
arg = sys.argv[4] if len(sys.argv) > 4 else 'auto'
if arg == 'auto':
    starts = [r[0] for r in rows]
    cum = [0]
    for s_, e_, _ in rows:
        cum.append(cum[-1] + e_ - s_)
    best, w0 = -1, t0
    cand = t0
    while cand + wlen <= t1:
        i0, i1 = bisect.bisect_left(starts, cand), bisect.bisect_left(starts, cand + wlen)
        if cum[i1] - cum[i0] > best:
            best, w0 = cum[i1] - cum[i0], cand
        cand += ms_step * 1e6 / 4
else:
    w0 = t0 + (t1 - t0) * float(arg)
w1 = w0 + wlen
win = [r for r in rows if r[0] >= w0 and r[1] <= w1]
busy = sum(e - s for s, e, _ in win)
span = win[-1][1] - win[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
overlap = 0
last_end, last_name = win[0][1], win[0][2]
for s, e, name in win[1:]:
    g = s - last_end
    if g > 0:
        gaps[last_name][0] += g
        gaps[last_name][1] += 1
    else:
        overlap += min(-g, e - s)
    if e > last_end:
        last_end, last_name = e, name
idle = sum(v[0] for v in gaps.values())
print('window: %d launches over %.2f ms (= %.2f steps of %.2f ms): kernel time %.2f ms (%.1f%%), idle %.2f ms (%.1f%%), overlapped %.2f ms'
      % (len(win), span / 1e6, span / 1e6 / ms_step, ms_step, busy / 1e6, 100.0 * busy / span, idle / 1e6, 100.0 * idle / span, overlap / 1e6))
print('per step: %.0f launches, kernel %.2f ms, idle %.2f ms' % (len(win) / (span / 1e6 / ms_step), busy / 1e6 / (span / 1e6 / ms_step),
                                                               idle / 1e6 / (span / 1e6 / ms_step)))
print('-- idle time by the kernel it follows')
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:20]:
    print('%7.3f ms  %6d gaps  avg %6.2f us  after %s' % (v[0] / 1e6, v[1], v[0] / v[1] / 1e3, k))
big = sorted(((s2 - e1, n1) for (s1, e1, n1), (s2, e2, n2) in zip(win, win[1:])), reverse=True)[:10]
print('-- largest gaps (us): ' + ', '.join('%.0f after %s' % (g / 1e3, nm[:24]) for g, nm in big))

byk = collections.defaultdict(lambda: [0, 0])
for s_, e_, name in win:
    byk[name][0] += e_ - s_
    byk[name][1] += 1
big_idle = sum(g for g, _ in big if g > 1e6)
steps = (span - big_idle) / 1e6 / ms_step
print('-- kernel time per step (window = %.2f steps after removing gaps > 1 ms)' % steps)
for k, v in sorted(byk.items(), key=lambda kv: -kv[1][0])[:45]:
    print('%7.3f ms/step %7.1f launches/step  avg %7.1f us  %s' % (v[0] / 1e6 / steps, v[1] / steps, v[0] / v[1] / 1e3, k))
