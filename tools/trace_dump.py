"""Analysis: print consecutive kernels of a rocprofv3 kernel_trace.csv with gaps: python tools/trace_dump.py <csv> <count> [skip_from_end]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]); skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows))
ev = ev[len(ev) - skip - n: len(ev) - skip]
t0 = ev[0][0]; prev_end = ev[0][0]
for s, e, name, q in ev:
    nm = re.sub(r'dcp::|void ', '', name)
    nm = re.sub(r'TileCfg<([^>]*)>', r'T<\1>', nm)[:110]
    print('%10.1f us  gap %8.1f  dur %8.1f  q%s  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, q, nm))
    prev_end = max(prev_end, e)
