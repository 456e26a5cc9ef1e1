"""Analysis: repeat golden dictionary-learning cases through decomp_amd.dictionary_learning.solve."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decomp_amd import dictionary_learning as dl
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'dl_golden.npz'))
names = [n for n in sorted(set(k.rsplit('/', 1)[0] for k in g.files)) if n.count('/') == 4 and '/nomask/' in n]
sel = [n for n in names if any(a in n for a in sys.argv[1:])] if len(sys.argv) > 1 else names
for rep in range(int(os.environ.get('REPS', '1'))):
    out = []
    for name in sel:
        parts = name.split('/')
        y, D0 = g[parts[0] + '/y'], g[parts[0] + '/D0']
        mb = int(parts[1][2:]); lm = parts[2].rstrip('0123456789'); li = int(parts[2][len(lm):]); ep = int(parts[4][2:])
        it, D, x = dl.solve(y.copy(), D0.copy(), 0.1, tol=0.0, minibatch=mb, maxiter=ep + 1, lasso_method=lm,
                            lasso_iter=li, lasso_tol=1e-5, random_seed=0)
        e = float(np.max(np.abs(D - g[name + '/D'])) / np.max(np.abs(g[name + '/D'])))
        ex = float(np.max(np.abs(x - g[name + '/x'])) / np.max(np.abs(g[name + '/x'])))
        out.append((name, '%.1e' % e, '%.1e' % ex))
    print('rep', rep, {k: os.environ.get(k) for k in ('DCP_DL_PREFETCH', 'DCP_DL_SYNC')}, flush=True)
    for o in out:
        print('   ', o, flush=True)
