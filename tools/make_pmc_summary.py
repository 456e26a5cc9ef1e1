#!/usr/bin/env python3
"""Per-kernel means of every counter found under <src>/pmc_*/ (rocprofv3 --pmc, one pass per directory), with the
HBM traffic corrected as MI355X_MICROARCH.md prescribes.  Usage: make_pmc_summary.py <src_dir> <out.json>"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    return re.sub(r'\(.*', '', name).replace('void ', '').replace('dcp::', '')


src, out = sys.argv[1], sys.argv[2]
pmc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for path in glob.glob(os.path.join(src, 'pmc_*', '*', '*_counter_collection.csv')):
    seen = set()
    for r in csv.DictReader(open(path)):
        if 'dcp::' not in r['Kernel_Name'] and 'move_rows' not in r['Kernel_Name']:
            continue
        k = short(r['Kernel_Name'])
        pmc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        key = r['Dispatch_Id']
        if key not in seen:
            seen.add(key)
            dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
summary = {}
for k, cs in pmc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d['dispatches_profiled'] = len(dur[k])
    d['avg_us_profiled'] = sum(dur[k]) / len(dur[k])
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        kmajor_stream = ('gemm_mfma_kernel' in k) and (', 0, 0, ' in k or ', 0, 1, ' in k)
        factor = 1.10 if kmajor_stream else 2.0      # profiles/r01_fetch_calibration.txt
        d['fetch_correction_factor'] = factor
        d['hbm_bytes_guide_x2'] = (2.0 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024.0
        d['hbm_bytes_corrected'] = (factor * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024.0
    if 'GRBM_GUI_ACTIVE' in d and 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
        cyc = d['GRBM_GUI_ACTIVE'] / 8.0
        d['mfma_pipe_util'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc
    if 'SQ_WAVE_CYCLES' in d and 'SQ_WAIT_INST_ANY' in d:
        d['wait_inst_frac_of_wave_cycles'] = d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']
    summary[k] = d
json.dump(summary, open(out, 'w'), indent=1, sort_keys=True)
for k in sorted(summary, key=lambda k: -summary[k]['avg_us_profiled'] * summary[k]['dispatches_profiled'])[:12]:
    d = summary[k]
    print('%-110s n=%3d %8.1f us  mfma_util %.3f  wait_inst %.3f  hbm %.3g B' % (
        k[:110], d['dispatches_profiled'], d['avg_us_profiled'], d.get('mfma_pipe_util', float('nan')),
        d.get('wait_inst_frac_of_wave_cycles', float('nan')), d.get('hbm_bytes_corrected', float('nan'))))
