import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import decomp_amd
from decomp_amd import _arrays, _hip
for (N, F, K, dt) in [(256, 128, 8, np.float64), (256, 128, 8, np.float32), (2048, 512, 32, np.float32), (8192, 1024, 64, np.float32)]:
    rng = np.random.RandomState(0)
    xt = np.maximum(rng.randn(N, K), 0); Dt = np.maximum(rng.randn(K, F), 0)
    y = (xt @ Dt + 0.1 * np.abs(rng.randn(N, F))).astype(dt)
    D0 = np.maximum(Dt + 0.3 * rng.randn(K, F), 0.1).astype(dt)
    yd = torch.from_numpy(y).cuda(); Dd = torch.from_numpy(D0).cuda()
    decomp_amd.nmf.solve(yd, Dd.clone(), tol=0.0, maxiter=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it, D, x = decomp_amd.nmf.solve(yd, Dd.clone(), tol=0.0, maxiter=1001)
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    print('%5dx%4d k=%2d %s: %.1f us/iteration' % (N, F, K, np.dtype(dt).name, dt_s / 1000 * 1e6))
