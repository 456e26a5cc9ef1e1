#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (short name), mean of
each counter per dispatch and the dispatch duration.  Usage: pmc_summary.py <csv> [filter]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(.*', '', name)
    name = name.replace('void ', '').replace('dcp::', '')
    return name[:100]


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        if flt and flt not in k:
            continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for k in sorted(acc, key=lambda k: -sum(dur[k].values())):
        d = list(dur[k].values())
        print('%s\n   dispatches %d  avg_us %.1f  vgpr/lds n/a' % (k, len(d), sum(d) / len(d)))
        for c, v in sorted(acc[k].items()):
            print('   %-28s %.4g' % (c, sum(v) / len(v)))


if __name__ == '__main__':
    main()
