"""The reference's own LASSO property tests (tests/test_lasso.py:60-226, 283-476) re-expressed
against decomp_amd on the GPU, over EVERY solver of ``lasso.AVAILABLE_METHODS`` /
``AVAILABLE_NNLS_METHODS``: the returned x is a local minimum of the objective under random
perturbations, masks behave (all-ones == none, 1-D == tiled), all solvers agree with ista, badly
conditioned designs still converge, and the shape / dtype error surface."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Problem(object):
    """Data generators of tests/test_lasso.py (same seeds, same call order)."""
    tol = 1.0e-6

    def __init__(self, kind='real', shape=(), K=5, F=10, correlated=0.0, nnls=False, f32=False):
        self.rng = np.random.RandomState(0)
        self.kind, self.f32 = kind, f32
        if correlated:
            self.A = self.randn(K, F) + self.randn(F) * correlated
        else:
            self.A = self.randn(K, F)
        n = int(np.prod(shape)) if shape else 1
        if nnls:
            x_true = np.maximum(self.randn(K), 0.0)
        else:
            x_true = (self.randn(n * K) * np.rint(self.rng.uniform(size=n * K))).reshape(tuple(shape) + (K,))
        self.y = np.dot(x_true, self.A) + self.randn(*(tuple(shape) + (F,))) * 0.1
        self.mask = np.rint(self.rng.uniform(0.4, 1.0, size=n * F)).reshape(tuple(shape) + (F,))
        if f32:
            self.A, self.y = self.A.astype(np.float32), self.y.astype(np.float32)
            self.mask = self.mask.astype(np.float32)
        self.nnls = nnls

    def randn(self, *s):
        if self.kind == 'complex':
            return self.rng.randn(*s) + self.rng.randn(*s) * 1.0j
        return self.rng.randn(*s)

    def error(self, x, alpha, mask):                       # test_lasso.py:123-129, 283-291
        if self.nnls:
            x = np.maximum(x, 0.0)
        if mask is None:
            mask = np.ones(self.y.shape)
        a = alpha * np.sum(mask, axis=-1, keepdims=True)
        loss = np.sum(0.5 / a * np.square(np.abs(self.y - np.tensordot(x, self.A, axes=1))) * mask)
        return loss + np.sum(np.abs(x))

    def assert_minimum(self, x, alpha, tol, mask=None, n=100, msg=None):   # test_lasso.py:131-135
        loss = self.error(x, alpha, mask)
        for _ in range(n):
            dx = self.randn(*x.shape) * tol
            assert loss <= self.error(x + dx, alpha, mask) * (1.0 + self.tol), msg


def _methods(nnls=False):
    from decomp_amd import lasso
    return list(lasso.AVAILABLE_NNLS_METHODS if nnls else lasso.AVAILABLE_METHODS)


CASES = {
    'vector': dict(), 'matrix': dict(shape=(11,)), 'tensor': dict(shape=(12, 11)),
    'complex': dict(kind='complex'), 'complex_matrix': dict(kind='complex', shape=(11,)),
    'complex_tensor': dict(kind='complex', shape=(12, 11)),
    'matrix_float32': dict(shape=(11,), f32=True),
    'nnls': dict(nnls=True),
}


@pytest.mark.parametrize('case', sorted(CASES))
def test_minimum_and_masks(case):
    """test_lasso.py:160-226 (TestLasso.test / test_mask / test_mask1d and its subclasses)."""
    from decomp_amd import lasso
    p = Problem(**CASES[case])
    single = p.f32
    tol = 1.0e-5 if single else p.tol
    for alpha in ((0.01, 0.1) if p.nnls else (0.1, 1.0)):       # test_lasso.py:197, 302
        for method in _methods(p.nnls):
            msg = '%s %s alpha %g' % (case, method, alpha)
            it, x = lasso.solve(p.y, p.A, alpha=alpha, tol=tol, method=method, maxiter=1000)
            assert it < 1000 - 1, msg
            p.assert_minimum(x, alpha, tol=tol, msg=msg)
            assert not np.allclose(x, 0.0), msg
            # with a mask
            mtol = 1.0e-4 if case == 'complex_tensor' else tol      # test_lasso.py:266-268
            it, xm = lasso.solve(p.y, p.A, alpha=alpha, tol=mtol, method=method, maxiter=1000, mask=p.mask)
            assert it < 1000 - 1, msg
            p.assert_minimum(xm, alpha, tol=mtol, mask=p.mask, msg=msg)
            # an all-ones mask is no mask
            it, xo = lasso.solve(p.y, p.A, alpha=alpha, tol=tol, method=method, maxiter=1000,
                                 mask=np.ones(p.mask.shape, dtype=p.mask.dtype))
            assert np.allclose(xo, x, atol=max(tol, 1e-6) * (10 if single else 1)), msg


@pytest.mark.parametrize('case', ['vector', 'matrix', 'tensor', 'complex_matrix', 'nnls'])
def test_mask_1d_equals_tiled(case):
    """test_lasso.py:205-226."""
    from decomp_amd import lasso
    p = Problem(**CASES[case])
    for alpha in ((0.001, 0.01) if p.nnls else (0.01, 0.1)):    # test_lasso.py:223, 312
        for method in _methods(p.nnls):
            msg = '%s %s alpha %g' % (case, method, alpha)
            m1 = np.rint(p.rng.uniform(0.4, 1.0, size=p.y.shape[-1]))
            mt = np.ones(p.y.shape) * m1
            it, x = lasso.solve(p.y, p.A, alpha=alpha, tol=p.tol, method=method, maxiter=1000, mask=m1)
            assert it < 1000 - 1, msg
            p.assert_minimum(x, alpha, tol=p.tol * 10.0, mask=mt, msg=msg)
            assert not np.allclose(x, 0.0), msg
            it, xt = lasso.solve(p.y, p.A, alpha=alpha, tol=p.tol, method=method, maxiter=1000, mask=mt)
            assert np.allclose(x, xt, atol=p.tol * 10, rtol=1.0e-3), msg


@pytest.mark.parametrize('variant', ['real', 'complex', 'real_illconditioned', 'complex_alpha1'])
def test_all_solvers_reach_the_ista_solution(variant):
    """test_lasso.py:318-411: TestLasso_equivalence, _complex, _illcondition (8 atoms x 5
    channels, alpha 0.5) and _illcondition_complex -- which, by its base class, is the
    well-conditioned complex design at alpha 1.0."""
    from decomp_amd import lasso
    if variant == 'real_illconditioned':
        p = Problem(kind='real', shape=(10,), K=8, F=5)
        alpha = 0.5
    else:
        p = Problem(kind='real' if variant == 'real' else 'complex', shape=(11,))
        alpha = 1.0 if variant == 'complex_alpha1' else 0.1
    _, x0 = lasso.solve(p.y, p.A, alpha=alpha, tol=1.0e-6, method='ista', maxiter=1000)
    _, xm0 = lasso.solve(p.y, p.A, alpha=alpha, tol=1.0e-6, method='ista', maxiter=1000, mask=p.mask)
    for method in _methods():
        if method == 'ista':
            continue
        it, x = lasso.solve(p.y, p.A, alpha=alpha, tol=1.0e-6, method=method, maxiter=1000)
        assert it < 1000 - 1, method
        if method != 'fista':
            assert np.allclose(x - x0, 0.0, atol=1.0e-4), method
        assert not np.allclose(x, 0.0), method
        it, xm = lasso.solve(p.y, p.A, alpha=alpha, tol=1.0e-6, method=method, maxiter=1000, mask=p.mask)
        assert it < 1000 - 1, method
        if method != 'fista':
            assert np.allclose(xm - xm0, 0.0, atol=1.0e-4), method


@pytest.mark.parametrize('K', [9, 5])
def test_highly_correlated_design(K):
    """test_lasso.py:423-476 (TestLasso_bad_condition, _bad_condition2)."""
    from decomp_amd import lasso
    p = Problem(shape=(11,), K=K, correlated=0.3)
    for method in _methods():
        for alpha in np.exp(np.linspace(np.log(0.1), np.log(10.0), 3)):
            msg = '%s alpha %g' % (method, alpha)
            it, x = lasso.solve(p.y, p.A, alpha=alpha, tol=1.0e-6, method=method, maxiter=3000)
            assert it < 3000 - 1, msg
            p.assert_minimum(x, alpha, tol=1.0e-5, msg=msg)
            it, x = lasso.solve(p.y, p.A, alpha=alpha, tol=1.0e-6, method=method, maxiter=3000, mask=p.mask)
            assert it < 3000 - 1, msg
            p.assert_minimum(x, alpha, tol=1.0e-5, mask=p.mask, msg=msg)


def test_error_surface():
    """test_lasso.py:60-120: shape and dtype errors for vector, matrix and tensor inputs."""
    from decomp_amd import lasso
    from decomp_amd.utils.exceptions import ShapeMismatchError, DtypeMismatchError
    r = np.random.randn
    u = np.random.uniform
    for y, A, x, mask in [
            (r(5), r(3, 4), None, None), (r(5), r(3, 5), r(4), None), (r(5), r(3, 5), r(3), u(0, 1, 4)),
            (r(2, 5), r(3, 4), None, None), (r(2, 5), r(3, 5), r(1, 3), None),
            (r(2, 5), r(3, 5), r(2, 3), u(0, 1, 8).reshape(2, 4)),
            (r(2, 4, 5), r(3, 4), None, None), (r(2, 4, 5), r(3, 5), r(2, 3, 3), None),
            (r(2, 4, 5), r(3, 5), r(2, 4, 3), u(0, 1, 24).reshape(2, 4, 3))]:
        with pytest.raises(ShapeMismatchError):
            lasso.solve(y, A, alpha=1.0, x=x, mask=mask)
    with pytest.raises(DtypeMismatchError):
        lasso.solve(r(5).astype(float), r(3, 4).astype(complex), alpha=1.0)
    with pytest.raises(DtypeMismatchError):
        lasso.solve(r(5), r(3, 4), alpha=1.0, mask=r(3, 4).astype(int))
    with pytest.raises(DtypeMismatchError):
        lasso.solve(r(5), r(3, 4), alpha=1.0, mask=r(3, 4).astype(complex))
    with pytest.raises(DtypeMismatchError):
        lasso.solve(r(5).astype(np.float32), r(3, 4).astype(np.float64), alpha=1.0)
