"""GPU parity: decomp_amd.lasso (HIP, through the C ABI) against the golden vectors of the
real reference (every starred method x {real, complex, float32} x {no mask, 1-D, 2-D mask}
x {vector, matrix, tensor}) and against the CPU oracle at larger sizes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _g():
    return np.load(os.path.join(GOLDEN, 'lasso_golden.npz'), allow_pickle=False)


def _err(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(1.0, float(np.max(np.abs(b))))


@pytest.mark.parametrize('kind', ['f64', 'c128', 'f32'])
@pytest.mark.parametrize('sname', ['vec', 'mat', 'ten'])
@pytest.mark.parametrize('mname', ['nomask', 'mask1d', 'mask2d'])
def test_golden_cases(kind, sname, mname):
    from decomp_amd import lasso
    g = _g()
    base = 'lasso_%s_%s' % (kind, sname)
    y, A = g[base + '/y'], g[base + '/A']
    mask = None if mname == 'nomask' else g[base + '/' + mname]
    methods = ['ista', 'acc_ista', 'fista', 'cd']
    if kind != 'c128':
        methods += [m + '_pos' for m in methods]
    for method in methods:
        for tag in ('conv', 'exh'):
            name = '%s/%s/%s/%s' % (base, mname, method, tag)
            it, x = lasso.solve(y.copy(), A.copy(), float(g[name + '/alpha']),
                                tol=float(g[name + '/tol']), method=method,
                                maxiter=int(g[name + '/maxiter']),
                                mask=None if mask is None else mask.copy())
            xr = g[name + '/x']
            assert x.shape == xr.shape and x.dtype == y.dtype, name
            if kind == 'f32':
                # float32: the solution within 2e-4 of the largest entry; the iteration count exact
                # on exhaustion and within ONE stop check (the test fires every 10th iteration,
                # lasso.py:293) on converged cases, where the last |dx| < tol decision is made on
                # single-precision rounding noise
                assert _err(x, xr) < 2e-4, (name, _err(x, xr))
                if tag == 'exh':
                    assert it == int(g[name + '/it']), name
                else:
                    assert abs(it - int(g[name + '/it'])) <= 10, (name, it, int(g[name + '/it']))
            else:
                assert it == int(g[name + '/it']), (name, it, int(g[name + '/it']))
                assert _err(x, xr) < 1e-8, (name, _err(x, xr))


def test_prox_known_answers_public_helpers():
    """tests/test_lasso.py:15-56 hand values on the public helper functions."""
    from decomp_amd import lasso
    assert np.allclose(lasso.soft_threshold_float(np.array([0.1, -2.0, 1.4]), 1.0), [0.0, -1.0, 0.4])
    z = np.array([0.1, -2.0, 1.4])
    assert np.allclose(lasso.soft_threshold_complex(z * 1.0j, 1.0),
                       lasso.soft_threshold_complex(z + 0j, 1.0) * 1.0j)
    assert np.allclose(lasso.soft_threshold_positive(z, 1.0), [0.0, 0.0, 0.4])


@pytest.mark.parametrize('method', ['ista', 'acc_ista', 'fista', 'cd', 'ista_pos', 'cd_pos'])
@pytest.mark.parametrize('dt', ['float32', 'float64', 'complex64'])
def test_against_oracle_medium(method, dt):
    """2048 x 384, K = 192 (exercises the MFMA path for float32, multi-tile generic path
    otherwise): 30 iterations, no early stop, same iterate as the oracle."""
    from decomp_amd import lasso
    from oracle import lasso as olasso
    if dt == 'complex64' and method.endswith('_pos'):
        pytest.skip('positive solvers are real only')
    rng = np.random.RandomState(3)
    N, F, K = 2048, 384, 192
    if method.startswith('cd'):      # the oracle's as-written CD costs K times more: keep it small
        N, F, K = 300, 96, 80
    cplx = dt == 'complex64'

    def randn(*s):
        return (rng.randn(*s) + 1j * rng.randn(*s)) if cplx else rng.randn(*s)
    A = randn(K, F)
    xt = randn(N, K) * (rng.uniform(size=(N, K)) < 0.05)
    y = xt @ A + 0.1 * randn(N, F)
    A, y = A.astype(dt), y.astype(dt)
    alpha = 0.05
    it, x = lasso.solve(y, A, alpha, tol=1e-9, method=method, maxiter=30)
    ito, xo = olasso.solve(y.copy(), A.copy(), alpha, tol=1e-9, method=method, maxiter=30)
    if method.startswith('cd') and dt != 'float64':
        # the Gram-form sweep can reach a bit-exact fixed point (all dx == 0) in fp32 where the
        # reference's recomputed-residual form keeps jittering by an ulp: it stops at a check
        assert it in (0, 10, 20, 29), it
    else:
        assert it == ito
    tol = 2e-4 if dt != 'float64' else 1e-9
    assert _err(x, xo) < tol, _err(x, xo)
    assert np.count_nonzero(x) > 0


def test_mask_equivalences():
    """tests/test_lasso.py:185-226: all-ones mask == no mask; 1-D mask == tiled 2-D mask."""
    from decomp_amd import lasso
    rng = np.random.RandomState(0)
    A = rng.randn(5, 10)
    xt = (rng.randn(55) * np.rint(rng.uniform(size=55))).reshape(11, 5)
    y = xt @ A + rng.randn(11, 10) * 0.1
    for method in ['ista', 'acc_ista', 'fista', 'cd']:
        _, x0 = lasso.solve(y, A, alpha=0.1, tol=1e-6, method=method, maxiter=1000)
        _, x1 = lasso.solve(y, A, alpha=0.1, tol=1e-6, method=method, maxiter=1000,
                            mask=np.ones(y.shape))
        assert np.allclose(x0, x1, atol=1e-6), method
        m1 = np.rint(rng.uniform(0.4, 1.0, size=10))
        it, xa = lasso.solve(y, A, alpha=0.1, tol=1e-6, method=method, maxiter=1000, mask=m1)
        assert it < 999
        _, xb = lasso.solve(y, A, alpha=0.1, tol=1e-6, method=method, maxiter=1000,
                            mask=np.ones(y.shape) * m1)
        assert np.allclose(xa, xb, atol=1e-5, rtol=1e-3), method


def test_errors():
    from decomp_amd import lasso
    y, A = np.random.randn(4, 6), np.random.randn(3, 6)
    with pytest.raises(ValueError):            # unknown method (lasso.py:88-90)
        lasso.solve(y, A, 0.1, method='newton')
    with pytest.raises(ValueError):            # nnls default method quirk (nnls.py:4-7)
        from decomp_amd import nnls
        nnls.solve(y, A, 0.1)
    with pytest.raises(AssertionError):        # negative mask (lasso.py:78)
        lasso.solve(y, A, 0.1, mask=-np.ones((4, 6)))
    with pytest.raises(AssertionError):        # complex + _pos (lasso.py:92)
        lasso.solve(y.astype(complex), A.astype(complex), 0.1, method='ista_pos')
    # maxiter exhaustion returns maxiter - 1
    it, x = lasso.solve(y, A, 0.1, tol=0.0, maxiter=13, method='fista')
    assert it == 12


@pytest.mark.parametrize('method', ['ista', 'acc_ista', 'fista', 'ista_pos'])
@pytest.mark.parametrize('mask_kind', ['mask2d', 'mask1d'])
def test_masked_against_oracle_medium_fp32(method, mask_kind):
    """float32 with a mask at tile-aligned sizes (2048 x 384, K = 192), i.e. through the fast-path
    kernels and their 16-byte epilogues: the chained (v A o M) A^H product with the per-row threshold
    scale (2-D mask, lasso.py:306-328) and the folded 1-D mask, against the oracle."""
    from decomp_amd import lasso
    from oracle import lasso as olasso
    rng = np.random.RandomState(11)
    N, F, K = 2048, 384, 192
    A = rng.randn(K, F).astype(np.float32)
    xt = (rng.randn(N, K) * (rng.uniform(size=(N, K)) < 0.05))
    if method.endswith('_pos'):
        xt = np.abs(xt)
    y = (xt @ A + 0.1 * rng.randn(N, F)).astype(np.float32)
    if mask_kind == 'mask2d':
        mask = (rng.uniform(size=(N, F)) > 0.25).astype(np.float32)
    else:
        mask = (rng.uniform(size=F) > 0.25).astype(np.float32)
    it, x = lasso.solve(y, A, 0.05, tol=1e-9, method=method, maxiter=25, mask=mask.copy())
    ito, xo = olasso.solve(y.copy(), A.copy(), 0.05, tol=1e-9, method=method, maxiter=25, mask=mask.copy())
    assert it == ito
    assert _err(x, xo) < 2e-4, _err(x, xo)
    assert np.count_nonzero(x) > 0


@pytest.mark.parametrize('dt', ['float32', 'float64', 'complex64', 'complex128'])
def test_random_shapes_against_oracle(dt):
    """Seeded random (samples, channels, atoms) off every tile grid, all gradient methods and masks of
    every rank, 20 iterations without early stop: the same iterate as the oracle (lasso.py restated)."""
    from decomp_amd import lasso
    from oracle import lasso as olasso
    cplx = dt.startswith('complex')
    rng = np.random.RandomState({'float32': 11, 'float64': 12, 'complex64': 13, 'complex128': 14}[dt])
    tol = 3e-4 if dt in ('float32', 'complex64') else 1e-9
    methods = ['ista', 'acc_ista', 'fista'] + ([] if cplx else ['ista_pos', 'fista_pos'])
    for trial in range(14):
        N = int(rng.choice([1, 7, 33, 100, 257, 1000, 2049]))
        F = int(rng.choice([5, 30, 64, 100, 130, 250, 517]))
        K = int(rng.choice([3, 8, 20, 33, 64, 100, 129, 200]))
        method = methods[trial % len(methods)]
        mkind = trial % 3

        def randn(*s):
            return (rng.randn(*s) + 1j * rng.randn(*s)) if cplx else rng.randn(*s)
        A = randn(K, F)
        xt = randn(N, K) * (rng.uniform(size=(N, K)) < 0.2)
        y = (xt @ A + 0.1 * randn(N, F)).astype(dt)
        A = A.astype(dt)
        mask = None
        if mkind == 1:
            mask = np.rint(rng.uniform(0.3, 1.0, size=F)).astype(y.real.dtype)
        elif mkind == 2:
            mask = np.rint(rng.uniform(0.3, 1.0, size=(N, F))).astype(y.real.dtype)
        it, x = lasso.solve(y.copy(), A.copy(), 0.05, tol=1e-12, method=method, maxiter=20,
                            mask=None if mask is None else mask.copy())
        ito, xo = olasso.solve(y.copy(), A.copy(), 0.05, tol=1e-12, method=method, maxiter=20,
                               mask=None if mask is None else mask.copy())
        assert x.shape == xo.shape and x.dtype == y.dtype
        e = _err(x, xo)
        assert e < tol, (dt, trial, N, F, K, method, mkind, e)


@pytest.mark.parametrize('dt,K', [('float64', 2112), ('float32', 2112), ('complex128', 2112), ('float64', 4160)])
def test_coordinate_descent_wider_than_2048_atoms(dt, K):
    """The reference's coordinate descent has no width limit (lasso.py:526-552); beyond the 2048 atoms a
    wave's registers hold, the sweep keeps a row's x and g in memory (cd_gram_wide_kernel).  Against the
    oracle's as-written sweep; method 'cd' and 'parallel_cd' (whose p <= 1 fallback is cd: with K > F the
    Gershgorin bound exceeds K / 2)."""
    import decomp_amd as decomp
    from oracle import lasso as olasso
    rng = np.random.RandomState(K)
    cplx = dt.startswith('complex')
    N, F = 9, 48

    def randn(*s):
        return (rng.randn(*s) + 1j * rng.randn(*s)) if cplx else rng.randn(*s)
    A = randn(K, F).astype(dt)
    xt = (randn(N, K) * (rng.uniform(size=(N, K)) < 0.01)).astype(dt)
    y = (xt @ A + 0.05 * randn(N, F)).astype(dt)
    single = dt == 'float32'
    for method in ('cd', 'parallel_cd'):
        ito, xo = olasso.solve(y.copy(), A.copy(), 0.05, tol=1e-4, method=method, maxiter=12)
        it, x = decomp.lasso.solve(y.copy(), A.copy(), 0.05, tol=1e-4, method=method, maxiter=12)
        assert x.dtype == y.dtype and x.shape == (N, K)
        scale = max(1.0, float(np.max(np.abs(xo))))
        if single:
            assert abs(it - ito) <= 10 and np.max(np.abs(x - xo)) < 2e-3 * scale
        else:
            assert it == ito and np.max(np.abs(x - xo)) < 1e-8 * scale, (method, it, ito)


def test_wide_coordinate_descent_equals_register_form_bitwise(monkeypatch):
    """cd_gram_wide_kernel does the arithmetic of cd_gram_kernel in the same order: with the register form's
    limit lowered through DCP_CD_REGISTER_LIMIT (a test knob) a 1500-atom problem takes the wide form and
    must reproduce the register form's codes BIT FOR BIT in every dtype.  (Round 3 had to loosen the complex
    case: the compiler contracted the complex multiply-adds of the two kernels differently; both kernels now
    form g + x a and g - dx a with explicit fused multiply-adds in a fixed order, scalar.hpp fmadd / fmsub.)"""
    import decomp_amd as decomp
    rng = np.random.RandomState(3)
    for dt in ('float32', 'complex64', 'float64', 'complex128'):
        cplx = dt.startswith('complex')
        N, F, K = 37, 96, 1500

        def randn(*s):
            return (rng.randn(*s) + 1j * rng.randn(*s)) if cplx else rng.randn(*s)
        A = randn(K, F).astype(dt)
        y = ((randn(N, K) * (rng.uniform(size=(N, K)) < 0.02)) @ A + 0.05 * randn(N, F)).astype(dt)
        monkeypatch.delenv('DCP_CD_REGISTER_LIMIT', raising=False)
        it_a, x_a = decomp.lasso.solve(y.copy(), A.copy(), 0.05, tol=1e-5, method='cd', maxiter=21)
        monkeypatch.setenv('DCP_CD_REGISTER_LIMIT', '1024')
        it_b, x_b = decomp.lasso.solve(y.copy(), A.copy(), 0.05, tol=1e-5, method='cd', maxiter=21)
        assert it_a == it_b, dt
        assert np.array_equal(x_a, x_b), dt
        assert np.count_nonzero(x_a) > 0


@pytest.mark.parametrize('dt', ['float64', 'float32'])
def test_coordinate_descent_exit_at_sweep_0_and_10_and_warm_start(dt):
    """Ten coordinate-descent sweeps = the check sweep in a launch of its own + up to nine more in a second launch
    that the device skips when the check sweep met the test (lasso.py:546-551; rounds 3-4 parked the post-check codes
    in a snapshot instead).  Pins the exit at sweep 0, a later check (exit at sweep 10 / 20) and the warm-start
    regime -- a second solve from the converged codes exits at sweep 0 -- against the oracle's as-written sweep:
    iteration counts identical, codes to rounding."""
    import decomp_amd as decomp
    from oracle import lasso as olasso
    rng = np.random.RandomState(12)
    N, F, K = 50, 40, 24
    A = rng.randn(K, F).astype(dt)
    y = ((rng.randn(N, K) * (rng.uniform(size=(N, K)) < 0.2)) @ A + 0.05 * rng.randn(N, F)).astype(dt)
    tol = 1e-3 if dt == 'float32' else 1e-6
    eps = 2e-4 if dt == 'float32' else 1e-9
    it1, x1 = decomp.lasso.solve(y.copy(), A.copy(), 0.05, tol=tol, method='cd', maxiter=200)
    ito, xo = olasso.solve(y.copy(), A.copy(), 0.05, tol=tol, method='cd', maxiter=200)
    assert it1 == ito and it1 >= 10 and it1 % 10 == 0                 # met at a LATER check sweep
    assert np.max(np.abs(x1 - xo)) < eps * max(1.0, np.max(np.abs(xo)))
    for _ in range(2):      # warm starts: the nine sweeps behind the check sweep are skipped on the device
        it2, x2 = decomp.lasso.solve(y.copy(), A.copy(), 0.05, x=x1.copy(), tol=tol, method='cd', maxiter=200)
        it2o, x2o = olasso.solve(y.copy(), A.copy(), 0.05, x=xo.copy(), tol=tol, method='cd', maxiter=200)
        assert it2 == it2o == 0                                        # met at sweep 0
        assert np.max(np.abs(x2 - x2o)) < eps * max(1.0, np.max(np.abs(xo)))
    # a cold problem right after: the check sweep fails, nine more follow
    y3 = (y + 0.5 * rng.randn(N, F)).astype(dt)
    it3, x3 = decomp.lasso.solve(y3.copy(), A.copy(), 0.05, tol=tol, method='cd', maxiter=200)
    it3o, x3o = olasso.solve(y3.copy(), A.copy(), 0.05, tol=tol, method='cd', maxiter=200)
    assert it3 == it3o and it3 >= 10
    assert np.max(np.abs(x3 - x3o)) < eps * max(1.0, np.max(np.abs(x3o)))


@pytest.mark.parametrize('method', ['ista', 'acc_ista', 'fista'])
@pytest.mark.parametrize('maxiter', [1, 2, 10, 11, 12, 21])
def test_proximal_gradient_iteration_count_edges(method, maxiter):
    """Round 4 folded the solver's x * s / x / s passes into its first and last launches and reads the stop flag one
    iteration late: the edges of that bookkeeping -- one iteration (first = check = last), the iteration after a check,
    a check ON the last iteration (maxiter = 11, 21), exhaustion, and an early exit at iteration 0 or 10 -- against
    the oracle, iteration counts exact (float64)."""
    import decomp_amd as decomp
    from oracle import lasso as olasso
    rng = np.random.RandomState(maxiter)
    N, F, K = 70, 33, 17
    A = rng.randn(K, F)
    y = (rng.randn(N, K) * (rng.uniform(size=(N, K)) < 0.3)) @ A + 0.05 * rng.randn(N, F)
    for tol in (0.0, 1e-2, 1e3):          # never met / met at a later check / met at iteration 0
        it, x = decomp.lasso.solve(y.copy(), A.copy(), 0.05, tol=tol, method=method, maxiter=maxiter)
        ito, xo = olasso.solve(y.copy(), A.copy(), 0.05, tol=tol, method=method, maxiter=maxiter)
        assert it == ito, (method, maxiter, tol, it, ito)
        assert np.max(np.abs(x - xo)) < 1e-9 * max(1.0, np.max(np.abs(xo))), (method, maxiter, tol)
