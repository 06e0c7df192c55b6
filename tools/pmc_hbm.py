"""profiles/r03_pmc_hbm.json from two rocprofv3 PMC passes over tools/hbm_micro.py (FETCH_SIZE and WRITE_SIZE in separate passes,
MI355X_MICROARCH.md; units KB summed over the TCC instances; FETCH_SIZE may under-count streaming reads by up to 2x on gfx950)."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].split('(')[0]
        agg[name][0] += float(r['Counter_Value'])
        agg[name][1].add(r['Dispatch_Id'])
    return {k: (v[0] / len(v[1]), len(v[1])) for k, v in agg.items()}


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
want = ('adam_multi_k', 'pool_boxsum_k', 'pool_unbox_k', 'pack_weight_kernel', 'bn_', 'conv_igemm_kernel', 'conv_stem_kernel', 'conv_thin_kernel')
kern = {}
for k in sorted(set(fetch) | set(write)):
    if any(t in k for t in want):
        kern[k] = {'FETCH_SIZE_KB_per_launch': fetch.get(k, (None, 0))[0], 'WRITE_SIZE_KB_per_launch': write.get(k, (None, 0))[0],
                   'launches': fetch.get(k, write.get(k))[1]}
print(json.dumps({'workload': 'tools/hbm_micro.py (txt2vid_amd.util.roofline.hbm_bound_lines): the HBM-bound kernels at the benchmark shapes',
                  'notes': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, KB per launch; 2 x FETCH is the upper bound on gfx950 '
                           '(guide: streaming reads may be under-counted by up to 2x); compare with bench.py hbm_bound.*.algorithmic_bytes',
                  'kernels': kern}, indent=1))
