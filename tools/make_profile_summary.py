#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into committed summaries:
    profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, dcp kernels + totals)
    profiles/<tag>_pmc_summary.json   (per-kernel PMC means; HBM traffic corrected per
                                       MI355X_MICROARCH.md: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024)
Usage: make_profile_summary.py gpurun_out/prof_<tag> <tag>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(.*', '', name)
    return name.replace('void ', '').replace('dcp::', '')


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, 'profiles')
    os.makedirs(out_dir, exist_ok=True)
    # kernel stats
    ks = sorted(glob.glob(os.path.join(src, 'kt', '*', '*_kernel_stats.csv')), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(out_dir, tag + '_kernel_stats.csv'), 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py   (default: --gpus 1 --steps 50 --warmup 5)\n')
        f.write('Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n')
        for r in rows:
            f.write('"%s",%s,%s,%s,%s,%s,%s\n' % (short(r['Name'])[:160], r['Calls'], r['TotalDurationNs'],
                                                  r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']))
    # PMC
    pmc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for path in glob.glob(os.path.join(src, 'pmc_*', '*', '*_counter_collection.csv')):
        seen = set()
        for r in csv.DictReader(open(path)):
            if 'dcp::' not in r['Kernel_Name']:
                continue
            k = short(r['Kernel_Name'])
            pmc[k][r['Counter_Name']].append(float(r['Counter_Value']))
            key = (path, r['Dispatch_Id'])
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    summary = {}
    for k, cs in pmc.items():
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d['avg_us_profiled'] = sum(dur[k]) / len(dur[k])
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            # gfx950: FETCH_SIZE under-counts wide coalesced reads by exactly 2 (guide); the
            # K-contiguous panel loader's 64-byte row segments were calibrated separately
            # (profiles/r01_fetch_calibration.txt): factor 1.10
            kmajor_stream = ('gemm_mfma_kernel' in k) and (', 0, 0, ' in k or ', 0, 1, ' in k)
            factor = 1.10 if kmajor_stream else 2.0
            d['fetch_correction_factor'] = factor
            d['hbm_bytes_guide_x2'] = (2.0 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024.0
            d['hbm_bytes_corrected'] = (factor * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024.0
        if 'GRBM_GUI_ACTIVE' in d and 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
            cyc = d['GRBM_GUI_ACTIVE'] / 8.0
            d['mfma_pipe_util'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc
        summary[k] = d
    with open(os.path.join(out_dir, tag + '_pmc_summary.json'), 'w') as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print('wrote', tag)


if __name__ == '__main__':
    main()
