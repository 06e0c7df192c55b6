"""Per-launch shape table of the convolution launches of an iteration from a T2V_PROF_DUMP csv (bench.py / t2v_prof_end):
python tools/launch_table.py <dump.csv> <iterations in the dump>. One line per distinct (kind, M, Cin, Cout, taps, members, S,
kernel): launches per iteration, ms per iteration, us per launch, GFLOP per launch, TFLOP/s."""
import collections
import csv
import sys

KINDS = {0: 'fwd/dgrad', 1: 'wgrad', 2: 'wgrad-reduce', 3: 'thin', 4: 'splitk-reduce', 5: 'bf16 fwd/dgrad'}
PLAN0 = {0: 'igemm', 1: 'strip', 2: 'thin', 3: 'linear', 4: 'thin2', 5: 'strip3', 6: 'igemm_bf16', 8: 'strip3_bf16', 9: 'pool_fwd', 10: 'pool_dgrad', 11: 'pool_wgrad', 12: 'stem'}
WPLAN0 = {0: 'wgrad taps', 1: 'wgrad cols', 2: 'wgrad rows3', 3: 'wgrad gemm', 11: 'pool_wgrad'}


def main(path, steps):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        plan = [int(v) for v in r['plan'].split(':')]
        kind = int(r['kind'])
        if kind == 1:
            kern = WPLAN0.get(plan[0], '?')
            tile = 'S=%d cps=%d' % (plan[1], plan[2])
        elif plan[0] < 0:
            kern, tile = KINDS.get(kind, '?'), ''
        else:
            kern = PLAN0.get(plan[0], '?')
            tile = '%dx%dx%d' % (plan[1], plan[2], plan[3])
        key = (KINDS.get(kind, str(kind)), kern, tile, int(r['M']), int(r['Cin']), int(r['Cout']), int(r['taps']), int(r['groups']), int(r['S']))
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(r['ms'])
        a[2] += float(r['flops'])
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    print('%-15s %-12s %-12s %8s %5s %5s %4s %3s %4s | %6s %8s %8s %8s %7s' % ('kind', 'kernel', 'tile', 'M', 'Cin', 'Cout', 'taps', 'mem', 'S', 'n/it',
                                                                                  'ms/it', 'us', 'GFLOP', 'TF/s'))
    for k, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tot[k[0]][0] += ms / steps
        tot[k[0]][1] += fl / steps
        tot[k[0]][2] += n / steps
        print('%-15s %-12s %-12s %8d %5d %5d %4d %3d %4d | %6.1f %8.3f %8.1f %8.2f %7.1f' % (k + (n / steps, ms / steps, ms / n * 1e3, fl / n / 1e9,
                                                                                             fl / ms / 1e9 if ms > 0 else 0.0)))
    print()
    for k, (ms, fl, n) in tot.items():
        print('%-15s %6.1f launches/it %8.3f ms/it %8.4f TFLOP/it %7.1f TFLOP/s' % (k, n, ms, fl / 1e12, fl / ms / 1e9 if ms > 0 else 0.0))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
