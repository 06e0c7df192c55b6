"""Histogram of the conv launches recorded by T2V_PROF_DUMP: time and achieved TFLOP/s per FLOP bucket."""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
for kind, name in ((0, 'igemm fwd/dgrad'), (1, 'wgrad')):
    b = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for r in rows:
        if int(r['kind']) != kind:
            continue
        fl, ms = float(r['flops']), float(r['ms'])
        e = 0
        while (1 << e) * 1e6 < fl:
            e += 1
        k = b[e]
        k[0] += fl; k[1] += ms; k[2] += 1
    tot = sum(v[1] for v in b.values())
    print('%s: total %.2f ms/step' % (name, tot / steps))
    for e in sorted(b):
        fl, ms, n = b[e]
        print('  <= %8.0f MFLOP: %5.0f launches/step  %6.2f ms/step (%4.1f%%)  %6.1f TFLOP/s  avg %6.1f us' %
              ((1 << e), n / steps, ms / steps, 100 * ms / tot, fl / ms / 1e9 if ms else 0, 1e3 * ms / n))
