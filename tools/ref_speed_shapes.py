"""The reference's OWN benchmark shapes (speed_tests/tests/test_nmf.py:14-66: NMF 'mu', F=500, N=2000, K=10, 100
iterations, with and without mask; speed_tests/tests/test_lasso.py:8-59: y 1100x100, A 90x100, float32, 5 alphas,
tol=1e-4, maxiter=5000, every method): decomp_amd on device arrays (the reference's `xp.array(...)` convention) beside
the NumPy oracle on the host cores.  Data recipes restated from those files; nothing is imported from them.
Usage: python tools/ref_speed_shapes.py [--no-cpu]; prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def nmf_data(F=500, N=2000, K=10):
    rng = np.random.RandomState(0)
    rn = lambda *s: rng.randn(*s).astype(np.float32)
    xt = np.maximum(rn(N, K), 0.0)
    Dt = np.maximum(rn(K, F), 0.0)
    y = np.dot(xt, Dt)
    D = Dt + 0.3 * np.maximum(rn(*Dt.shape), 0.)
    v = rng.uniform(0.45, 1.0, size=y.size).reshape(y.shape)
    return y, D, np.rint(v).astype(np.float32)


def lasso_data():
    rng = np.random.RandomState(0)
    A = rng.randn(90, 100) + rng.randn(100) * 0.3          # highly correlated design matrix
    xt = (rng.randn(99000) * np.rint(rng.uniform(size=99000))).reshape(1100, 90)
    y = np.dot(xt, A) + rng.randn(1100, 100) * 0.1
    v = rng.uniform(0.45, 1.0, size=110000).reshape(1100, 100)
    alphas = np.exp(np.linspace(np.log(0.1), np.log(10.0), 5)).astype(np.float32)
    return y.astype(np.float32), A.astype(np.float32), np.rint(v).astype(np.float32), alphas


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


def _blas_threads_within_quota():
    """Cap the host BLAS at the container's CPU quota (threads beyond it get the whole group throttled)."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q == 'max':
            return None, None
        n = max(1, int(float(q) / float(per)))
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=n, user_api='blas'), n
    except Exception:
        return None, None


def run(with_cpu=True, cpu_budget_s=20.0):
    import torch
    import decomp_amd
    from oracle import nmf as onmf, lasso as olasso
    limiter, nthreads = _blas_threads_within_quota()
    res = {'nmf_mu_F500_N2000_K10_100it': {}, 'lasso_1100x100_A90x100_5alphas': {},
           'cpu_blas_threads': nthreads if nthreads is not None else 'library default'}
    y, D, mask = nmf_data()
    yd, Dd, md = (torch.from_numpy(a).cuda() for a in (y, D, mask))
    for tag, m_np, m_dev in (('nomask', None, None), ('mask', mask, md)):
        def gpu():
            it, Do, xo = decomp_amd.nmf.solve(yd, Dd, tol=1.0e-10, method='mu', maxiter=100, mask=m_dev)
            torch.cuda.synchronize()
            return it
        s, it = timed(gpu, 5)
        e = {'gpu_ms_per_call': round(1e3 * s, 3), 'gpu_us_per_iteration': round(1e6 * s / 99, 2), 'it': int(it)}
        if with_cpu:
            sc, itc = timed(lambda: onmf.solve(y, D.copy(), tol=1.0e-10, maxiter=100, mask=m_np)[0], 2)
            e.update({'cpu_ms_per_call': round(1e3 * sc, 3), 'cpu_us_per_iteration': round(1e6 * sc / 99, 2),
                      'it_cpu': int(itc), 'gpu_speedup': round(sc / s, 2)})
        res['nmf_mu_F500_N2000_K10_100it'][tag] = e
    y, A, mask, alphas = lasso_data()
    yd, Ad = torch.from_numpy(y).cuda(), torch.from_numpy(A).cuda()
    spent = 0.0
    for method in decomp_amd.lasso.AVAILABLE_METHODS:
        def gpu():
            its = []
            for a in alphas:
                it, x = decomp_amd.lasso.solve(yd, Ad, alpha=float(a), tol=1.0e-4, method=method, maxiter=5000)
                its.append(int(it))
            torch.cuda.synchronize()
            return its
        s, its = timed(gpu, 2)
        e = {'gpu_ms_per_call': round(1e3 * s, 2), 'it': its, 'gpu_us_per_iteration': round(1e6 * s / max(1, sum(i + 1 for i in its)), 2)}
        if with_cpu and spent < cpu_budget_s:
            t0 = time.perf_counter()
            itc = []
            for a in alphas:
                it, x = olasso.solve(y, A, float(a), tol=1.0e-4, method=method, maxiter=5000)
                itc.append(int(it))
            sc = time.perf_counter() - t0
            spent += sc
            e.update({'cpu_ms_per_call': round(1e3 * sc, 2), 'it_cpu': itc, 'gpu_speedup': round(sc / s, 2)})
        res['lasso_1100x100_A90x100_5alphas'][method] = e
    if limiter is not None:
        limiter.restore_original_limits()
    return res


if __name__ == '__main__':
    print(json.dumps(run(with_cpu='--no-cpu' not in sys.argv)))
