set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/small_stats
rm -rf $O; mkdir -p $O
cat > $O/run.py <<'PY'
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import decomp_amd
N, F, K = [int(v) for v in sys.argv[1:4]]
rng = np.random.RandomState(0)
xt = np.maximum(rng.randn(N, K), 0); Dt = np.maximum(rng.randn(K, F), 0)
y = (xt @ Dt + 0.1 * np.abs(rng.randn(N, F))).astype(np.float32)
D0 = np.maximum(Dt + 0.3 * rng.randn(K, F), 0.1).astype(np.float32)
decomp_amd.nmf.solve(torch.from_numpy(y).cuda(), torch.from_numpy(D0).cuda(), tol=0.0, maxiter=201)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $O/run.py "$@" > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/small_stats/kt/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print('%6d calls %9.1f us avg %6.2f %%  %s' % (int(r['Calls']), float(r['AverageNs'])/1e3, float(r['Percentage']), r['Name'][:120]))
PY
