"""The reference's NMF property tests (tests/test_nmf.py:14-266) re-expressed against decomp_amd on
the GPU: the full-batch multiplicative update ends in a local minimum of the loss (l2 / kl, with and
without mask), every stochastic variant keeps decreasing the loss, and the "lazy transfer" runs (host
y streamed to a device D -- CuPy-only in the reference) behave the same."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _l2_strict(D):
    return D / np.sqrt(np.sum(D * D, axis=-1, keepdims=True))


class Problem(object):
    def __init__(self, likelihood, n_samples, mask_low):
        self.rng = np.random.RandomState(0)
        self.likelihood = likelihood
        self.Dtrue = np.maximum(self.rng.randn(3, 20), 0.0)                # test_nmf.py:62-69
        self.xtrue = np.maximum(self.rng.randn(n_samples, 3), 0.0)
        ytrue = self.xtrue @ self.Dtrue
        noise = self.rng.randn(*ytrue.shape) * 0.1
        self.y = ytrue + (np.abs(noise) if likelihood == 'kl' else noise)  # :21-29
        self.D = np.maximum(self.Dtrue + self.rng.randn(*self.Dtrue.shape) * 0.3, 0.1)
        self.mask = np.rint(self.rng.uniform(mask_low, 1, size=self.y.size)).reshape(self.y.shape)

    def error(self, x, D, mask):                                           # :31-46
        mask = np.ones(self.y.shape) if mask is None else mask
        D = _l2_strict(np.asarray(D))
        if self.likelihood == 'l2':
            return 0.5 * np.sum(np.square(self.y - x @ D) * mask)
        f = np.maximum(x @ D, 1.0e-15)
        return np.sum((-self.y * np.log(f) + f) * mask)

    def assert_minimum(self, x, D, tol, n=100, mask=None):                 # :48-54
        loss = self.error(x, D, mask)
        for _ in range(n):
            xn = np.maximum(x + self.rng.randn(*x.shape) * tol, 0.0)
            Dn = np.maximum(D + self.rng.randn(*D.shape) * tol, 0.0)
            assert loss < self.error(xn, Dn, mask) + 1.0e-15


@pytest.mark.parametrize('likelihood', ['l2', 'kl'])
@pytest.mark.parametrize('masked', [False, True])
def test_fullbatch_reaches_a_minimum(likelihood, masked):
    """test_nmf.py:60-102 (TestFullbatch_L2 / _KL, test_run and test_run_mask)."""
    from decomp_amd import nmf
    p = Problem(likelihood, 101, 0.3)
    mask = p.mask if masked else None
    it, D, x = nmf.solve(p.y, p.D.copy(), x=None, tol=1.0e-6, minibatch=None, maxiter=3000,
                         method='mu', likelihood=likelihood, mask=mask, random_seed=0)
    assert it < 3000 - 1
    p.assert_minimum(x, D, tol=1.0e-5, n=100, mask=mask)
    assert not np.allclose(x, 0.0, atol=1.0e-5)


def _decreasing_loss(p, method, likelihood, mask, lazy):
    """test_nmf.py:122-152 (_run_minibatch)."""
    import torch
    from decomp_amd import nmf

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def host(a):
        return a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    D0 = dev(p.D.copy()) if lazy else p.D.copy()       # lazy transfer: device D, host y / mask
    kw = dict(tol=1.0e-6, minibatch=30, method=method, likelihood=likelihood, mask=mask, random_seed=0)
    it, D, x = nmf.solve(p.y, D0, x=None, maxiter=31, **kw)
    start = p.error(host(x), host(D), mask)
    assert not np.allclose(host(x), 0.0, atol=1.0e-5)
    errors = []
    for _ in range(10):
        it, D, x = nmf.solve(p.y, D, x=x, maxiter=31, **kw)
        e = p.error(host(x), host(D), mask)
        assert not np.allclose(host(x), 0.0, atol=1.0e-5)
        assert e < start
        errors.append(e)
    errors = np.array(errors)
    assert np.mean(errors[:4]) > np.mean(errors[-4:])


@pytest.mark.parametrize('method', ['svrmu', 'svrmu-acc', 'asg-mu', 'gsg-mu', 'asag-mu', 'gsag-mu'])
@pytest.mark.parametrize('likelihood', ['l2', 'kl'])
@pytest.mark.parametrize('masked', [False, True])
def test_minibatch_variants_keep_decreasing_the_loss(method, likelihood, masked):
    """test_nmf.py:159-266 (Test_SVRMU_* ... Test_GSAG_MU_*, test_run and test_run_mask)."""
    p = Problem(likelihood, 1001, 0.375)
    _decreasing_loss(p, method, likelihood, p.mask if masked else None, lazy=False)


@pytest.mark.parametrize('method', ['svrmu', 'asag-mu'])
@pytest.mark.parametrize('masked', [False, True])
def test_minibatch_lazy_transfer(method, masked):
    """test_nmf.py:172-178 (test_run_lazy_transfer(_mask); CuPy-only in the reference): host y and
    mask are streamed through AsyncMinibatchData to a device-resident D."""
    p = Problem('l2', 1001, 0.375)
    _decreasing_loss(p, method, 'l2', p.mask if masked else None, lazy=True)
