"""Analysis: split a rocprofv3 kernel_trace.csv of tools/dl_e2e.py into solve() calls (idle gaps > 250 us) and print,
per call, first-kernel-to-last-kernel time, kernel count, busy time and the largest inner gaps.
python tools/e2e_calls.py <kernel_trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
calls, cur, prev_end = [], [], None
for s, e, n in ev:
    if prev_end is not None and s - prev_end > 250e3:
        calls.append(cur); cur = []
    cur.append((s, e, n)); prev_end = max(prev_end or 0, e)
calls.append(cur)
print('pieces:', [len(c) for c in calls])
for i in range(1, len(calls)):
    a, b = calls[i - 1], calls[i]
    print('idle %.1f us between piece %d (last: %s) and piece %d (first: %s)' % ((b[0][0] - max(e for _, e, _ in a)) / 1e3, i - 1, re.sub(r'dcp::|void ', '', a[-1][2])[:40], i, re.sub(r'dcp::|void ', '', b[0][2])[:40]))
for c in calls[-4:]:
    t0, t1 = c[0][0], max(e for _, e, _ in c)
    busy, pe, gaps = 0, c[0][0], []
    for s, e, n in c:
        if s > pe:
            gaps.append(((s - pe) / 1e3, (s - t0) / 1e3, re.sub(r'dcp::|void ', '', n)[:50]))
        busy += max(0, e - max(s, pe)); pe = max(pe, e)
    gaps.sort(reverse=True)
    print('call: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms; largest gaps (us, at us, before kernel):' %
          (len(c), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    for g in gaps[:8]:
        print('    %8.1f us at %10.1f  %s' % g)
