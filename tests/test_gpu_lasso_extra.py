"""GPU parity for the two remaining LASSO solvers of the reference, parallel_cd and admm
(lasso.py:448-523, 586-657): HIP path through the C ABI against golden
vectors of the real reference and against the CPU oracle at a size that runs on the MFMA tiles."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _g():
    return np.load(os.path.join(GOLDEN, 'lasso_extra_golden.npz'), allow_pickle=False)


def _err(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(1.0, float(np.max(np.abs(b))))


def _run_case(g, name):
    from decomp_amd import lasso
    base, mname, method, tag = name.split('/')
    y, A = g[base + '/y'], g[base + '/A']
    mask = None if mname == 'nomask' else g[base + '/' + mname]
    kw = dict(tol=float(g[name + '/tol']), method=method, maxiter=int(g[name + '/maxiter']),
              mask=None if mask is None else mask.copy())
    if str(g[name + '/raises']) == 'TypeError':     # lasso.py:509
        with pytest.raises(TypeError):
            lasso.solve(y.copy(), A.copy(), float(g[name + '/alpha']), **kw)
        return
    it, x = lasso.solve(y.copy(), A.copy(), float(g[name + '/alpha']), **kw)
    xr = g[name + '/x']
    # admm: the reference hands back double for single-precision input (promotion by
    # `rho * eye(K)`, lasso.py:603); this build keeps the problem dtype.
    assert x.shape == xr.shape and x.dtype == y.dtype, name
    single = y.dtype in (np.float32, np.complex64)
    if single:
        assert _err(x, xr) < 1e-3, (name, _err(x, xr))
        if tag == 'exh':
            assert it == int(g[name + '/it']), name
    else:
        assert it == int(g[name + '/it']), (name, it, int(g[name + '/it']))
        assert _err(x, xr) < 1e-8, (name, _err(x, xr))


@pytest.mark.parametrize('kind', ['f64', 'c128', 'f32'])
@pytest.mark.parametrize('sname', ['vec', 'mat', 'ten'])
def test_golden_reference_test_shapes(kind, sname):
    g = _g()
    prefix = 'lasso_%s_%s/' % (kind, sname)
    names = [str(c) for c in g['cases'] if str(c).startswith(prefix)]
    assert len(names) == (12 if kind == 'c128' else 24)
    for name in names:
        _run_case(g, name)


@pytest.mark.parametrize('kind', ['f64', 'f32', 'c128', 'c64'])
@pytest.mark.parametrize('tag', ['wide', 'corr'])
def test_golden_wide_and_fallback(kind, tag):
    """24 atoms x 64 channels: several coordinates committed per parallel_cd iteration
    ('wide'); nearly parallel atoms give p <= 1, i.e. the fallback to plain cd and, with a
    full mask, the reference's TypeError ('corr')."""
    g = _g()
    prefix = 'lasso_%s_%s/' % (kind, tag)
    names = [str(c) for c in g['cases'] if str(c).startswith(prefix)]
    assert names
    for name in names:
        _run_case(g, name)


@pytest.mark.parametrize('method', ['parallel_cd', 'admm', 'parallel_cd_pos', 'admm_pos'])
@pytest.mark.parametrize('dt', ['float32', 'float64', 'complex64'])
@pytest.mark.parametrize('masked', [False, True])
def test_against_oracle_medium(method, dt, masked):
    """1024 x 256, K = 128: MFMA tiles for float32 / complex64, 25 iterations without early
    stop, same iterate as the oracle (the oracle's shuffle stream is the reference's)."""
    from decomp_amd import lasso
    from oracle import lasso as olasso
    if dt == 'complex64' and method.endswith('_pos'):
        pytest.skip('positive solvers are real only')
    rng = np.random.RandomState(5)
    N, F, K = 1024, 256, 128
    if masked and method.startswith('admm'):
        N, F, K = 96, 64, 24      # one K x K inverse per row
    cplx = dt == 'complex64'

    def randn(*s):
        return (rng.randn(*s) + 1j * rng.randn(*s)) if cplx else rng.randn(*s)
    A = randn(K, F)
    xt = randn(N, K) * (rng.uniform(size=(N, K)) < 0.05)
    y = (xt @ A + 0.1 * randn(N, F)).astype(dt)
    A = A.astype(dt)
    mask = None
    if masked:
        mask = np.rint(rng.uniform(0.4, 1.0, size=(N, F))).astype(np.float64 if dt == 'float64'
                                                                   else np.float32)
    it, x = lasso.solve(y, A, 0.05, tol=1e-12, method=method, maxiter=25, mask=mask)
    ito, xo = olasso.solve(y.copy(), A.copy(), 0.05, tol=1e-12, method=method, maxiter=25,
                           mask=None if mask is None else mask.copy())
    assert it == ito == 24
    assert x.dtype == y.dtype
    tol = 2e-4 if dt != 'float64' else 1e-9
    assert _err(x, xo) < tol, _err(x, xo)
    assert np.count_nonzero(x) > 0


def test_admm_rho_keyword_and_nnls_wrapper():
    """lasso.solve(..., method='admm', rho=...) forwards rho (lasso.py:155) and
    nnls.solve appends '_pos' (nnls.py:4-7)."""
    from decomp_amd import lasso, nnls
    from oracle import lasso as olasso
    rng = np.random.RandomState(1)
    A = rng.randn(12, 40)
    y = np.abs(rng.randn(30, 12)) @ A + 0.1 * rng.randn(30, 40)
    it, x = lasso.solve(y, A, 0.1, tol=1e-8, method='admm', maxiter=500, rho=2.0)
    # the oracle's admm takes rho through its private entry point
    s = np.sqrt(np.sum(A * A, axis=-1))
    ito, xo = olasso._admm(y, A / s[:, None], 0.1 / s * A.shape[1], np.zeros((30, 12)) * s,
                           1e-8 * s, 500, False, None, rho=2.0)
    assert it == ito and _err(x, xo / s) < 1e-8
    it2, x2 = nnls.solve(y, A, 0.1, tol=1e-8, method='admm', maxiter=500)
    it3, x3 = lasso.solve(y, A, 0.1, tol=1e-8, method='admm_pos', maxiter=500)
    # admm returns the unconstrained iterate x (not z): non-negative only up to the tolerance
    assert it2 == it3 and np.array_equal(x2, x3) and np.all(x2 > -1e-6)


def test_masked_admm_system_larger_than_64_kib_of_lds():
    """with a 2-D mask ADMM solves one K x K system per row (lasso.py:620-657); the batched Gauss-Jordan
    and the masked step keep a row's pivot data in LDS, raised up to the CU's 160 KiB (the header advertises
    K <= 10240 real / 5120 complex).  K = 2112 complex128 needs > 64 KiB: two rows (71 MB each) against the oracle."""
    from decomp_amd import lasso
    from oracle import lasso as olasso
    rng = np.random.RandomState(8)
    N, F, K = 2, 48, 2112

    def randn(*s):
        return rng.randn(*s) + 1j * rng.randn(*s)
    A = randn(K, F).astype(np.complex128)
    xt = randn(N, K) * (rng.uniform(size=(N, K)) < 0.01)
    y = (xt @ A + 0.05 * randn(N, F)).astype(np.complex128)
    mask = np.rint(rng.uniform(0.4, 1.0, size=(N, F)))
    it, x = lasso.solve(y.copy(), A.copy(), 0.05, tol=1e-12, method='admm', maxiter=6, mask=mask.copy())
    ito, xo = olasso.solve(y.copy(), A.copy(), 0.05, tol=1e-12, method='admm', maxiter=6, mask=mask.copy())
    assert it == ito == 5
    assert x.dtype == y.dtype and x.shape == (N, K)
    assert _err(x, xo) < 1e-8, _err(x, xo)
