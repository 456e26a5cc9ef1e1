set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/dl_stats
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/bench_dl.py "$@" > $O/log.txt 2>&1
tail -1 $O/log.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/dl_stats/kt/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print('%8d calls %10.1f us avg %6.2f %%  %s' % (int(r['Calls']), float(r['AverageNs'])/1e3, float(r['Percentage']), r['Name'][:110]))
PY
