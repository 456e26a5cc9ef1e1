"""Analysis: timeline of a rocprofv3 kernel trace (kernel_trace.csv): busy time, idle gaps, per-kernel totals over the
last `frac` of the trace.  Usage: python tools/e2e_timeline.py <kernel_trace.csv> [frac=0.25] [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows))
t_end = ev[-1][1]
t_begin = ev[0][0]
cut = t_end - frac * (t_end - t_begin)
ev = [e for e in ev if e[0] >= cut]
span = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
gaps = []
last_name = ev[0][2]
for s, e, n, q in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name[:70], n[:70]))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = n
busy += cur_e - cur_s
print('window %.3f ms, %d kernels, busy %.3f ms (%.1f %%), idle %.3f ms; per step (/%d): span %.4f busy %.4f'
      % (span / 1e6, len(ev), busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, steps, span / 1e6 / steps, busy / 1e6 / steps))
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n, q in ev:
    tot[n[:90]] += e - s; cnt[n[:90]] += 1
print('--- kernel totals (sum of durations, overlaps counted twice) ---')
for n, t in tot.most_common(25):
    print('%9.3f ms %6d  %s' % (t / 1e6, cnt[n], n))
print('--- largest idle gaps ---')
agg = collections.Counter(); aggc = collections.Counter()
for g, a, b in gaps:
    agg[(a, b)] += g; aggc[(a, b)] += 1
for (a, b), g in agg.most_common(15):
    print('%9.3f ms %5d  after %s | before %s' % (g / 1e6, aggc[(a, b)], a, b))
queues = collections.Counter(q for _, _, _, q in ev)
print('queues', dict(queues))
