set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/stream_trace
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/bench_stream.py --dl-streamed-only > $O/log.txt 2>&1
tail -2 $O/log.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/stream_trace/kt/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Queue_Id','')) for r in rows]
ev.sort()
t0 = ev[0][0]
mid = len(ev)*2//3
w0 = ev[mid][0]
for s_,e,n,q in ev:
    if s_ < w0 or s_ > w0 + 25e6: continue
    if (e-s_) > 150e3 or 'move_rows' in n:
        print('%10.1f %8.1f q%s %s' % ((s_-t0)/1e3, (e-s_)/1e3, q, n[:50]))
PY
