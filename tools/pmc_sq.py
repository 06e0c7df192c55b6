"""Per-kernel SQ counters of a rocprofv3 --pmc pass (mean per launch) + derived MFMA utilisation.
  python tools/pmc_sq.py <rocprof dir>      MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
  (GRBM_GUI_ACTIVE comes back summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES = 64 cycles x the number of 32x32x2 f32 MFMAs)"""
import collections, csv, glob, json, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'].split('(')[0]
    if not any(t in name for t in ('conv_igemm', 'conv_wgrad', 'conv_pool')):
        continue
    agg[name][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[name].add(r['Dispatch_Id'])
out = {}
for k, v in agg.items():
    n = len(cnt[k])
    d = {c: val / n for c, val in v.items()}
    d['launches'] = n
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d and d['GRBM_GUI_ACTIVE'] > 0:
        d['MfmaUtil_percent'] = 100.0 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8.0 * 256 * 4)
    if 'SQ_WAVE_CYCLES' in d and d['SQ_WAVE_CYCLES'] > 0:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
            if c in d:
                d[c + '_share_of_wave_cycles'] = d[c] / d['SQ_WAVE_CYCLES']
    out[k] = d
print(json.dumps(out, indent=1))
