cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ista_prof
rm -rf $O; mkdir -p $O
cat > /tmp/ista_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
from decomp_amd import lasso
g = torch.Generator(device='cuda'); g.manual_seed(0)
N, F, K = 8192, 4096, 512
A = torch.randn((K, F), generator=g, device='cuda')
xt = 30.0 * torch.randn((N, K), generator=g, device='cuda') * (torch.rand((N, K), generator=g, device='cuda') < 0.05)
y = xt @ A + 0.1 * torch.randn((N, F), generator=g, device='cuda')
for _ in range(6):
    it, x = lasso.solve(y, A, 0.1, x=torch.ones((N, K), device='cuda'), tol=1e-5, method='ista', maxiter=10)
torch.cuda.synchronize()
print('it', it, float((x != 0).float().mean()))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python3 /tmp/ista_run.py > $O/p1.log 2>&1
DCP_ISTA_PERSIST=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p0 -- python3 /tmp/ista_run.py > $O/p0.log 2>&1
for d in p1 p0; do echo == $d; f=$(find $O/$d -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(r['Name'][:110], r['Calls'], '%.1f us' % (float(r['AverageNs']) / 1e3), r['Percentage'])
PY
done
