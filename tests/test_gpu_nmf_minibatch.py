"""GPU parity: the stochastic MU variants of decomp_amd.nmf.solve(minibatch=...) against
the golden vectors of the real reference (tests/test_nmf.py:105-152 shapes)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
METHODS = ['asg-mu', 'gsg-mu', 'asag-mu', 'gsag-mu', 'svrmu', 'svrmu-acc']


def _g():
    return np.load(os.path.join(GOLDEN, 'nmf_minibatch_golden.npz'), allow_pickle=False)


def _err(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) / \
        max(1e-300, float(np.max(np.abs(b))))


@pytest.mark.parametrize('method', METHODS)
@pytest.mark.parametrize('base', ['nmfmb_float64_l2', 'nmfmb_float64_kl', 'nmfmb_float32_l2'])
def test_golden(base, method):
    import decomp_amd
    g = _g()
    y, D0, mask = g[base + '/y'], g[base + '/D0'], g[base + '/mask']
    lik = 'kl' if base.endswith('_kl') else 'l2'
    f64 = y.dtype == np.float64
    for mtag in ('nomask', 'mask'):
        name = '%s/%s/%s/it3' % (base, method, mtag)
        it, D, x = decomp_amd.nmf.solve(y.copy(), D0.copy(), x=None, tol=0.0, minibatch=30,
                                        maxiter=3, method=method, likelihood=lik,
                                        mask=mask.copy() if mtag == 'mask' else None,
                                        random_seed=0)
        assert it == int(g[name + '/it']) and D.dtype == y.dtype and x.shape == (1001, 3)
        tol = 1e-8 if f64 else 1e-3
        assert _err(D, g[name + '/D']) < tol, (name, _err(D, g[name + '/D']))
        assert _err(x, g[name + '/x']) < tol, (name, _err(x, g[name + '/x']))
    name = '%s/%s/conv' % (base, method)
    it, D, x = decomp_amd.nmf.solve(y.copy(), D0.copy(), x=None, tol=3.0e-2, minibatch=30,
                                    maxiter=30, method=method, likelihood=lik, random_seed=0)
    if f64:
        assert it == int(g[name + '/it']), name
        assert _err(D, g[name + '/D']) < 1e-8 and _err(x, g[name + '/x']) < 1e-8, name


def test_kwargs_and_errors():
    import decomp_amd
    rng = np.random.RandomState(0)
    y, D = np.abs(rng.randn(200, 12)), np.abs(rng.randn(3, 12)) + 0.1
    a = decomp_amd.nmf.solve(y, D, tol=0.0, minibatch=20, maxiter=3, method='asag-mu',
                             forget_rate=0.3, random_seed=1)
    b = decomp_amd.nmf.solve(y, D, tol=0.0, minibatch=20, maxiter=3, method='asag-mu',
                             forget_rate=0.7, random_seed=1)
    assert not np.allclose(a[1], b[1])
    with pytest.raises(NotImplementedError):
        decomp_amd.nmf.solve(y, D, minibatch=20, method='nope')
    with pytest.raises(ValueError):
        decomp_amd.nmf.solve(y, D, minibatch=500, method='asg-mu')
