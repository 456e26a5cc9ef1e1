"""Per-step summary of a rocprofv3 kernel_stats.csv: python tools/trace_summary.py <csv> <steps> [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 18
tot = 0.0
for r in rows:
    if 'at::native' in r['Name'] or 'Cijk' in r['Name']:
        continue
    tot += float(r['TotalDurationNs']) / steps / 1e3
shown = 0
for r in rows:
    n = r['Name']
    if 'at::native' in n or 'Cijk' in n:
        continue
    n = re.sub(r'dcp::', '', n).replace('void ', '').split('(')[0][:104]
    print(f"{n:106s} calls/step {int(r['Calls']) / steps:6.1f}  us/step {float(r['TotalDurationNs']) / steps / 1e3:8.1f}  avg {float(r['AverageNs']) / 1e3:8.1f}")
    shown += 1
    if shown >= top:
        break
print('sum of kernel time per step: %.1f us' % tot)
