"""Analysis: slow HIP API calls in a rocprofv3 --hip-trace: python tools/hip_api_slow.py <hip_api_trace.csv> [min_us] [kernel_trace.csv]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
kn = {}
if len(sys.argv) > 3:
    for r in csv.DictReader(open(sys.argv[3])):
        kn[r['Correlation_Id']] = r['Kernel_Name']
tot = collections.Counter(); cnt = collections.Counter(); slow = collections.Counter(); slowt = collections.Counter()
t0 = min(int(r['Start_Timestamp']) for r in rows)
print('columns:', list(rows[0].keys()))
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Function']
    tot[n] += d; cnt[n] += 1
    if d >= thr:
        slow[n] += 1; slowt[n] += d
        print('slow %-28s at %10.3f ms  took %9.1f us  %s' % (n, (int(r['Start_Timestamp']) - t0) / 1e6, d,
                                                           kn.get(r.get('Correlation_Id', ''), '')[:90]))
print('function, calls, total ms, calls >= %.0f us, their total ms' % thr)
for n, t in tot.most_common(14):
    print('%-40s %7d %10.2f %7d %10.2f' % (n, cnt[n], t / 1e3, slow[n], slowt[n] / 1e3))
