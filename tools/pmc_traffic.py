"""Assemble profiles/r03_pmc_traffic.json from two rocprofv3 PMC passes over tools/conv_micro.py:

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python tools/conv_micro.py both 3
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python tools/conv_micro.py both 3
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r03_pmc_traffic.json

(one counter family per pass, MI355X_MICROARCH.md; FETCH_SIZE / WRITE_SIZE are in KB, summed over the TCC instances)."""
import collections
import csv
import glob
import json
import hashlib
import os
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
kern = {}
for k in sorted(set(fetch) | set(write)):
    if not any(t in k for t in ('conv_igemm', 'conv_wgrad', 'wgrad_reduce', 'conv_pool', 'pool_boxsum', 'pool_unbox')):
        continue
    kern[k] = {'FETCH_SIZE_KB_per_launch': fetch.get(k, (None, 0))[0], 'WRITE_SIZE_KB_per_launch': write.get(k, (None, 0))[0],
               'launches': fetch.get(k, write.get(k))[1]}
M, C, T = 49152, 64, 27
alg = (M * C * 4) * 2 + C * C * T * 4
Mfull, Mp = 393216, 49152
print(json.dumps({
    'workload': 'tools/conv_micro.py both: (1) DownBlock-0 first convolution 64->64 3x3x3 over the 8 discriminator-step members at B=32 '
                '(M = 49 152 voxels): forward / masked data gradient = conv_igemm_strip3_kernel<64,32,2,true> (the iteration\'s dominant kernel), '
                'weight gradient = conv_wgrad3_kernel + wgrad_reduce; (2) the stem\'s pooled second convolution 64->64 over its 8 members '
                '(393 216 voxels -> 49 152 pooled rows): pool_boxsum_k, conv_pool_fwd_kernel, conv_pool_dgrad_kernel, pool_unbox_k, conv_pool_wgrad_kernel',
    'conv_hip_sha1': hashlib.sha1(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'txt2vid_amd', 'csrc',
                                                   'conv.hip'), 'rb').read()).hexdigest(),
    'algorithmic_bytes_per_launch': {'strip3<64> fwd / dgrad (x + y + w)': alg, 'wgrad3 (x + gy + dw)': alg,
                                     'pool_boxsum (r + r~)': Mfull * C * 4 * 2, 'conv_pool_fwd (r~ + y + w)': Mfull * C * 4 + Mp * C * 4 + C * C * T * 4,
                                     'conv_pool_dgrad (gz + 8 planes over the 87 984 padded-grid voxels + w)': Mp * C * 4 + 8 * 87984 * C * 4 + C * C * T * 4,
                                     'pool_unbox (planes + mask + dr)': 8 * 87984 * C * 4 + 2 * Mfull * C * 4},
    'notes': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md: TCC slots). Units KB per launch. '
             'FETCH_SIZE may read low by up to 2x on gfx950 for streaming reads (guide), so 2 x FETCH is the upper bound.',
    'kernels': kern}, indent=1))
