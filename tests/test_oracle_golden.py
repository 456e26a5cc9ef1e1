"""The CPU oracle against the golden vectors produced by the real reference
(oracle/make_golden.py).  This is what pins the oracle: every later GPU parity
test compares the HIP path with this oracle."""
import os

import numpy as np
import pytest

import oracle
from oracle import nmf as onmf, lasso as olasso, dictionary_learning as odl


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _tol(dtype):
    # float64/complex128: same operations in the same order -> rounding only
    # float32: the reference promotes some intermediates (see oracle/lasso.py)
    return 1.0e-9 if np.dtype(dtype).itemsize >= 8 and np.dtype(dtype) != np.complex64 else 2.0e-4


def _close(a, b, tol):
    a = np.asarray(a)
    b = np.asarray(b)
    scale = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    return float(np.max(np.abs(a - b))) <= tol * scale if a.size else True


# ------------------------------------------------------------------ NMF ----
def _nmf_cases(golden_dir):
    g = _load(golden_dir, 'nmf_golden.npz')
    return g, [str(c) for c in g['cases']]


def test_nmf_trace_and_solve(golden_dir):
    g, cases = _nmf_cases(golden_dir)
    assert len(cases) == 24
    for name in cases:
        base, mtag = name.rsplit('_', 1)
        y, D0 = g[base + '/y'], g[base + '/D0']
        mask = g[base + '/mask'] if mtag == 'mask' else None
        lik = 'kl' if '_kl' in base else 'l2'
        tol = _tol(y.dtype)
        # per-iteration trace
        n = len(g[name + '/trace_maxdiff'])
        D = oracle.common.l2_strict(D0)
        x = np.ones((y.shape[0], D0.shape[0]), dtype=y.dtype)
        for i in range(n):
            x, D, diff = onmf.mu_step(y, x, D, mask, lik)
            ref = g[name + '/trace_maxdiff'][i]
            assert abs(diff - ref) <= tol * max(1.0, abs(ref)) + (1e-12 if tol < 1e-6 else 1e-6), (name, i)
            res = onmf.residual(y, x, D, mask)
            assert abs(res - g[name + '/trace_resid'][i]) <= max(tol, 1e-5 if y.dtype == np.float32 else 0) * g[name + '/trace_resid'][i] + 1e-12, (name, i)
        assert _close(D, g[name + '/trace_D'], tol), name
        assert _close(x, g[name + '/trace_x'], tol), name
        assert x.dtype == y.dtype and D.dtype == y.dtype
        # full solve
        it, Df, xf = onmf.solve(y, D0.copy(), tol=float(g[name + '/tol']),
                                maxiter=400, likelihood=lik, mask=mask)
        if y.dtype == np.float64:
            assert it == int(g[name + '/it']), name
            assert _close(Df, g[name + '/D'], tol), name
            assert _close(xf, g[name + '/x'], tol), name
        else:
            # float32: the stop iteration may move by rounding near tol
            assert abs(it - int(g[name + '/it'])) <= 3, name


def test_nmf_gram_formulation_drift(golden_dir):
    """The product computes x.(DD^T) and (x^T x).D; quantify the drift of that
    reformulation against the reference formulation (float32, 25 iterations):
    must stay far inside the 1e-5 residual tolerance of BASELINE.json."""
    g, _ = _nmf_cases(golden_dir)
    base = 'nmf_256x128k8_float32_l2'
    y, D0 = g[base + '/y'], g[base + '/D0']
    D = oracle.common.l2_strict(D0)
    x = np.ones((256, 8), np.float32)
    ref = g[base + '_nomask/trace_resid']
    for i in range(len(ref)):
        x, D, _ = onmf.mu_step_gram(y, x, D)
        assert abs(onmf.residual(y, x, D) - ref[i]) <= 1.0e-5 * ref[i]


# ---------------------------------------------------------------- LASSO ----
def test_lasso_prox_known_answers(golden_dir):
    g = _load(golden_dir, 'lasso_golden.npz')
    z = g['prox/z']
    # tests/test_lasso.py:21-33 hand values
    assert np.allclose(olasso.shrink_real(np.array([0.1, -2.0, 1.4]), 1.0),
                       [0.0, -1.0, 0.4])
    assert np.allclose(olasso.shrink_real(z, 1.0), [[0.0, -1.0, 0.4], [0.1, 2.0, -0.4]])
    assert np.allclose(olasso.shrink_real(z, 1.0), g['prox/real'])
    assert np.allclose(olasso.shrink_complex(z + z * 1.0j, 1.0), g['prox/complex45'])
    assert np.allclose(olasso.shrink_positive(z, 1.0), g['prox/positive'])
    # 90 degrees: shrinking i*z equals i*shrink(z)   (tests/test_lasso.py:42-45)
    zz = np.array([0.1, -2.0, 1.4])
    assert np.allclose(olasso.shrink_complex(zz * 1.0j, 1.0),
                       olasso.shrink_complex(zz + 0.0j, 1.0) * 1.0j)


def test_lasso_all_cases(golden_dir):
    g = _load(golden_dir, 'lasso_golden.npz')
    cases = [str(c) for c in g['cases']]
    assert len(cases) == 360
    for name in cases:
        base, mname, method, tag = name.split('/')
        y, A = g[base + '/y'], g[base + '/A']
        mask = None if mname == 'nomask' else g[base + '/' + mname]
        it, x = olasso.solve(y.copy(), A.copy(), float(g[name + '/alpha']),
                             tol=float(g[name + '/tol']), method=method,
                             maxiter=int(g[name + '/maxiter']),
                             mask=None if mask is None else mask.copy())
        xr = g[name + '/x']
        if y.dtype == np.float32:
            # reference promotes fista to float64 (NumPy>=2); compare loosely
            assert _close(x, xr, 5.0e-4), name
            if tag == 'exh':
                assert it == int(g[name + '/it']), name
        else:
            assert it == int(g[name + '/it']), name
            assert _close(x, xr, 1.0e-9), name
        assert x.shape == xr.shape


def test_lasso_parallel_cd_and_admm_cases(golden_dir):
    """lasso.py:448-523 and 586-657: the oracle performs the same
    operations on the same dtypes as the reference (including ADMM's silent promotion to
    double), so iteration counts and values agree to rounding for every dtype."""
    g = _load(golden_dir, 'lasso_extra_golden.npz')
    cases = [str(c) for c in g['cases']]
    assert len(cases) == 252
    n_raise = 0
    for name in cases:
        base, mname, method, tag = name.split('/')
        y, A = g[base + '/y'], g[base + '/A']
        mask = None if mname == 'nomask' else g[base + '/' + mname]
        kw = dict(tol=float(g[name + '/tol']), method=method, maxiter=int(g[name + '/maxiter']),
                  mask=None if mask is None else mask.copy())
        if str(g[name + '/raises']) == 'TypeError':
            n_raise += 1
            with pytest.raises(TypeError):
                olasso.solve(y.copy(), A.copy(), float(g[name + '/alpha']), **kw)
            continue
        it, x = olasso.solve(y.copy(), A.copy(), float(g[name + '/alpha']), **kw)
        xr = g[name + '/x']
        assert it == int(g[name + '/it']), name
        assert x.dtype == xr.dtype and x.shape == xr.shape, name
        assert _close(x, xr, 1.0e-9 if x.dtype.itemsize >= 8 and x.dtype != np.complex64
                      else 1.0e-5), name
    assert n_raise == 12


# ------------------------------------------------- dictionary learning -----
def test_dictionary_learning_all_cases(golden_dir):
    g = _load(golden_dir, 'dl_golden.npz')
    cases = [str(c) for c in g['cases']]
    assert len(cases) == 68
    for name in cases:
        parts = name.split('/')
        base = parts[0]
        y, D0, mask = g[base + '/y'], g[base + '/D0'], g[base + '/mask']
        if parts[1] == 'reftest':
            it, D, x = odl.solve(y.copy(), D0.copy(), 0.1, tol=1.0e-4,
                                 minibatch=100, maxiter=1000,
                                 lasso_method='acc_ista', lasso_iter=1000,
                                 random_seed=0)
        else:
            minibatch = int(parts[1][2:])
            lm = parts[2].rstrip('0123456789')
            li = int(parts[2][len(lm):])
            use_mask = parts[3] == 'mask'
            epochs = int(parts[4][2:])
            yy = y * mask if use_mask else y
            it, D, x = odl.solve(yy.copy(), D0.copy(), 0.1, tol=0.0,
                                 minibatch=minibatch, maxiter=epochs + 1,
                                 lasso_method=lm, lasso_iter=li,
                                 lasso_tol=1.0e-5, random_seed=0,
                                 mask=mask.copy() if use_mask else None)
        assert it == int(g[name + '/it']), name
        assert _close(D, g[name + '/D'], 1.0e-9), name
        assert _close(x, g[name + '/x'], 1.0e-9), name


def _run_dl_case(solve, g, name):
    """Re-run one case of dl_extra_golden.npz (names as oracle/make_golden.py::gen_dl_extra)."""
    parts = name.split('/')
    base = parts[0]
    y, D0 = g[base + '/y'], g[base + '/D0']
    if base.startswith('dlwide_'):
        lm = parts[1].rstrip('0123456789')
        li = int(parts[1][len(lm):])
        epochs = int(parts[2][2:])
        return solve(y.copy(), D0.copy(), 0.02, tol=0.0, minibatch=128, maxiter=epochs + 1,
                     lasso_method=lm, lasso_iter=li, lasso_tol=1.0e-5, random_seed=0)
    mask = g[base + '/mask']
    minibatch = int(parts[1][2:])
    lm = parts[2].rstrip('0123456789')
    li = int(parts[2][len(lm):])
    use_mask = parts[3] == 'mask'
    epochs = int(parts[4][2:])
    yy = y * mask if use_mask else y
    return solve(yy.copy(), D0.copy(), 0.1, tol=0.0, minibatch=minibatch, maxiter=epochs + 1,
                 lasso_method=lm, lasso_iter=li, lasso_tol=1.0e-5, random_seed=0,
                 mask=mask.copy() if use_mask else None)


def test_dictionary_learning_extra_cases(golden_dir):
    """float32 / complex64 runs of the reference's own test shapes and wide dictionaries
    (K = 160 / 80 atoms: several blocks of the product's blocked atom sweep)."""
    g = _load(golden_dir, 'dl_extra_golden.npz')
    cases = [str(c) for c in g['cases']]
    assert len(cases) == 48
    for name in cases:
        it, D, x = _run_dl_case(odl.solve, g, name)
        single = g[name + '/D'].dtype in (np.float32, np.complex64)
        assert D.dtype == g[name + '/D'].dtype and x.dtype == g[name + '/x'].dtype, name
        assert it == int(g[name + '/it']), name
        assert _close(D, g[name + '/D'], 2.0e-4 if single else 1.0e-9), name
        assert _close(x, g[name + '/x'], 2.0e-4 if single else 1.0e-9), name


# ------------------------------------------------------- stochastic MU NMF -----
def test_nmf_minibatch_all_cases(golden_dir):
    from oracle import nmf_minibatch as omb
    g = _load(golden_dir, 'nmf_minibatch_golden.npz')
    cases = [str(c) for c in g['cases']]
    assert len(cases) == 54
    for name in cases:
        parts = name.split('/')
        base, method = parts[0], parts[1]
        y, D0, mask = g[base + '/y'], g[base + '/D0'], g[base + '/mask']
        lik = 'kl' if base.endswith('_kl') else 'l2'
        if parts[2] == 'conv':
            it, D, x = omb.solve(y.copy(), D0.copy(), tol=3.0e-2, minibatch=30, maxiter=30,
                                 method=method, likelihood=lik, random_seed=0)
        else:
            it, D, x = omb.solve(y.copy(), D0.copy(), tol=0.0, minibatch=30,
                                 maxiter=int(parts[3][2:]), method=method, likelihood=lik,
                                 mask=mask.copy() if parts[2] == 'mask' else None, random_seed=0)
        tol = 1.0e-9 if y.dtype == np.float64 else 2.0e-4
        if y.dtype == np.float64 or parts[2] != 'conv':
            assert it == int(g[name + '/it']), name
            assert _close(D, g[name + '/D'], tol), name
            assert _close(x, g[name + '/x'], tol), name


def test_dictionary_minibatch_step_equals_solve():
    """oracle.dictionary_learning.minibatch_step (the one-step restatement the full-shape GPU gates and
    bench.py use) reproduces the per-step trace of oracle.dictionary_learning.solve -- which is pinned to the
    reference's fixtures above -- when driven with the same shuffles."""
    from oracle import dictionary_learning as odl
    from oracle.common import l2_strict
    rng = np.random.RandomState(5)
    for cplx in (False, True):
        y = rng.randn(60, 7) + (1j * rng.randn(60, 7) if cplx else 0)
        D0 = rng.randn(4, 7) + (1j * rng.randn(4, 7) if cplx else 0)
        trace = []
        odl.solve(y.copy(), D0.copy(), 0.1, tol=0.0, minibatch=20, maxiter=3, lasso_method='ista',
                  lasso_iter=10, lasso_tol=1e-5, random_seed=1, trace=trace)
        # replay: same RandomState stream, cumulative shuffles (dictionary_learning.py:131-133)
        r2 = np.random.RandomState(1)
        index = np.arange(60)
        ys, xs = y.copy(), np.ones((60, 4), dtype=D0.dtype)
        D = l2_strict(D0)
        A = np.zeros((4, 4), y.dtype)
        B = np.zeros((4, 7), y.dtype)
        count = 0
        for _ in range(2):
            r2.shuffle(index)
            ys, xs = ys[index], xs[index]
            for b in range(3):
                sl = slice(20 * b, 20 * b + 20)
                it2, xn, A, B, D, diff = odl.minibatch_step(ys[sl], xs[sl], D, A, B, count, 20, 0.1,
                                                            'ista', 10, 1e-5)
                xs[sl] = xn
                t = trace[count]
                assert it2 == t['lasso_it']
                assert np.array_equal(A, t['A']) and np.array_equal(B, t['B']) and np.array_equal(D, t['D'])
                assert diff == t['maxdiff']
                count += 1
