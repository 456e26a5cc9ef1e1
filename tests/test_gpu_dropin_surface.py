"""GPU parity of the public helper modules a caller of the reference imports directly
(decomp.utils.normalize, decomp.utils.cp_compat, decomp.math_utils.eigen) and of the
reference's one extension point, a user-supplied ``Likelihood`` (grads.py:12-13,17-93),
against the CPU oracle / the golden vectors of the real reference."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _rand(rng, shape, dtype):
    a = rng.randn(*shape)
    if np.dtype(dtype).kind == 'c':
        a = a + 1j * rng.randn(*shape)
    return a.astype(dtype)


@pytest.mark.parametrize('dtype', [np.float32, np.float64, np.complex64, np.complex128])
@pytest.mark.parametrize('strict', [True, False])
def test_normalize_matches_oracle(dtype, strict):
    """decomp/utils/normalize.py:2-21, incl. the strict=0 and complex entry points."""
    import decomp_amd as decomp
    from oracle import common
    rng = np.random.RandomState(3)
    tol = 2e-6 if np.dtype(dtype).itemsize in (4, 8) and np.dtype(dtype) in (np.float32, np.complex64) else 1e-13
    fn = decomp.utils.normalize.l2_strict if strict else decomp.utils.normalize.l2
    ref = common.l2_strict if strict else common.l2
    for shape in [(7, 33), (3, 5000), (1, 1), (2, 4, 19)]:
        U = _rand(rng, shape, dtype)
        if not strict:
            U[0] *= 0.01          # a row with |u|^2 < 1 must be left unscaled (max(.,1))
        keep = U.copy()
        out = fn(U, np)
        assert out.dtype == U.dtype and out.shape == U.shape and isinstance(out, np.ndarray)
        assert np.array_equal(U, keep)                          # input untouched
        assert np.max(np.abs(out - ref(U))) <= tol * max(1.0, float(np.max(np.abs(ref(U)))))
    # another axis (normalize.py takes axis=...): normalise the columns
    U = _rand(rng, (6, 9), dtype)
    out = fn(U, np, axis=0)
    assert np.max(np.abs(out - ref(U.T.copy()).T)) <= tol * 10


def test_normalize_torch_in_torch_out():
    import torch
    import decomp_amd as decomp
    U = torch.randn(5, 40, device='cuda', dtype=torch.float64)
    out = decomp.utils.normalize.l2_strict(U, None)
    assert out.is_cuda and out.data_ptr() != U.data_ptr()
    nrm = out.pow(2).sum(-1).sqrt().cpu().numpy()
    assert np.allclose(nrm, 1.0, atol=1e-13)


@pytest.mark.parametrize('dtype', [np.float32, np.float64, np.complex64, np.complex128])
def test_gershgorin_matches_oracle(dtype):
    """decomp/math_utils/eigen.py:9-20 on a matrix and on batches."""
    import decomp_amd as decomp
    from oracle import common
    rng = np.random.RandomState(4)
    for shape in [(5, 5), (4, 300, 300), (2, 3, 17, 17)]:
        X = _rand(rng, shape, dtype)
        got = decomp.math_utils.eigen.spectral_radius_Gershgorin(X, np)
        want = common.gershgorin(X)
        assert got.shape == want.shape == shape[:-2] + (1,)
        assert got.dtype == want.dtype
        assert np.max(np.abs(got - want)) <= (1e-4 if got.dtype == np.float32 else 1e-11) * np.max(want)


def test_cp_compat():
    import torch
    import decomp_amd as decomp
    cc = decomp.utils.cp_compat
    a, t = np.zeros(3), torch.zeros(3, device='cuda')
    assert cc.get_array_module(a, None, a) is np
    assert cc.get_array_module(t, None) is cc.numpy_or_cupy and cc.has_cupy
    with pytest.raises(TypeError, match='All the data types should be the same.'):
        cc.get_array_module(t, a)


def _problem(rng, N=101, F=20, K=3, dtype=np.float64):
    xt = np.maximum(rng.randn(N, K), 0)
    Dt = np.maximum(rng.randn(K, F), 0)
    y = (xt @ Dt + 0.1 * np.abs(rng.randn(N, F))).astype(dtype)
    D0 = np.maximum(Dt + 0.3 * rng.randn(K, F), 0.1).astype(dtype)
    mask = (rng.uniform(size=(N, F)) >= 0.2).astype(dtype)
    return y, D0, mask


def _user_gaussian(base):
    class Mine(base):
        """The reference's Gaussian (grads.py:108-125) restated as a user plugin, written with
        operators that both NumPy arrays and torch tensors support."""
        calls = 0

        def grad_x(self, y, x, d, mask):
            Mine.calls += 1
            f = x @ d
            if mask is not None:
                f = f * mask
                y = y * mask
            return y @ d.T, f @ d.T

        def grad_d(self, y, x, d, mask):
            f = x @ d
            if mask is not None:
                f = f * mask
                y = y * mask
            return x.T @ y, x.T @ f
    return Mine


@pytest.mark.parametrize('masked', [False, True])
def test_user_likelihood_numpy_matches_oracle_and_builtin(masked):
    """nmf.solve(likelihood=<Likelihood instance>) (grads.py:12-13): same iterates as the oracle
    and as the fused built-in 'l2' path."""
    import decomp_amd as decomp
    from oracle import nmf as onmf
    rng = np.random.RandomState(5)
    y, D0, mask = _problem(rng)
    m = mask if masked else None
    Mine = _user_gaussian(decomp.nmf_methods.grads.Likelihood)
    it, D, x = decomp.nmf.solve(y, D0.copy(), tol=1e-4, maxiter=60, likelihood=Mine(), mask=m)
    assert Mine.calls > 0 and isinstance(D, np.ndarray)
    ito, Do, xo = onmf.solve(y, D0.copy(), tol=1e-4, maxiter=60, mask=m)
    itb, Db, xb = decomp.nmf.solve(y, D0.copy(), tol=1e-4, maxiter=60, likelihood='l2', mask=m)
    assert it == ito == itb
    assert np.max(np.abs(D - Do)) < 1e-10 and np.max(np.abs(x - xo)) < 1e-9 * max(1.0, np.max(xo))
    assert np.max(np.abs(D - Db)) < 1e-8


def test_user_likelihood_torch_and_broadcast_parts():
    """torch CUDA in -> the plugin sees torch tensors; [1,K] / [K,1] parts broadcast as the
    reference's Poisson returns them (grads.py:146,155)."""
    import torch
    import decomp_amd as decomp
    from oracle import nmf as onmf
    rng = np.random.RandomState(6)
    y, D0, _ = _problem(rng)

    class Kl(decomp.nmf_methods.grads.Likelihood):
        seen = set()

        def grad_x(self, y, x, d, mask):
            Kl.seen.add(type(x).__name__)
            f = x @ d + 1.0e-15
            return (y / f) @ d.T, d.T.sum(0, keepdim=True)

        def grad_d(self, y, x, d, mask):
            f = x @ d + 1.0e-15
            return x.T @ (y / f), x.T.sum(1, keepdim=True)

    it, D, x = decomp.nmf.solve(torch.from_numpy(y).cuda(), torch.from_numpy(D0).cuda(), tol=0.0,
                                maxiter=15, likelihood=Kl())
    assert Kl.seen == {'Tensor'} and D.is_cuda and x.is_cuda
    ito, Do, xo = onmf.solve(y, D0.copy(), tol=0.0, maxiter=15, likelihood='kl')
    assert it == ito == 15
    assert np.max(np.abs(D.cpu().numpy() - Do)) < 1e-10


def test_user_likelihood_minibatch_matches_builtin():
    """A user Likelihood through the stochastic variants (serizel.py:36-60): identical to the
    fused kernels on the same shuffle."""
    import decomp_amd as decomp
    rng = np.random.RandomState(7)
    y, D0, mask = _problem(rng, N=300)
    Mine = _user_gaussian(decomp.nmf_methods.grads.Likelihood)
    for method in ('asg-mu', 'svrmu'):
        a = decomp.nmf.solve(y, D0.copy(), tol=0.0, minibatch=30, maxiter=4, method=method,
                             likelihood=Mine(), mask=mask, random_seed=2)
        b = decomp.nmf.solve(y, D0.copy(), tol=0.0, minibatch=30, maxiter=4, method=method,
                             likelihood='l2', mask=mask, random_seed=2)
        assert a[0] == b[0]
        assert np.max(np.abs(a[1] - b[1])) < 1e-9 and np.max(np.abs(a[2] - b[2])) < 1e-8 * max(1, np.max(b[2]))


def test_builtin_instances_and_unknown_likelihood():
    import decomp_amd as decomp
    g = decomp.nmf_methods.grads
    rng = np.random.RandomState(8)
    y, D0, _ = _problem(rng)
    a = decomp.nmf.solve(y, D0.copy(), tol=0.0, maxiter=5, likelihood=g.Gaussian())
    b = decomp.nmf.solve(y, D0.copy(), tol=0.0, maxiter=5, likelihood='l2')
    assert np.array_equal(a[1], b[1])
    assert isinstance(g.get_likelihood('kl'), g.Poisson)
    with pytest.raises(NotImplementedError):
        decomp.nmf.solve(y, D0.copy(), likelihood='nope')
    with pytest.raises(NotImplementedError):
        decomp.nmf.solve(y, D0.copy(), likelihood=object())


def test_svrmu_does_not_read_uninitialised_memory():
    """svrmu's P / Q start as torch.empty and are first written by axpby(b = 0).  Poison
    the caching allocator with NaN blocks of exactly those sizes, then require the golden result."""
    import torch
    import decomp_amd as decomp
    g = np.load(os.path.join(GOLDEN, 'nmf_minibatch_golden.npz'), allow_pickle=False)
    base = 'nmfmb_float64_l2'
    y, D0 = g[base + '/y'], g[base + '/D0']
    K, F = D0.shape
    for method in ('svrmu', 'svrmu-acc'):
        junk = [torch.full((K, F), float('nan'), dtype=torch.float64, device='cuda') for _ in range(24)]
        junk += [torch.full((K, F), float('inf'), dtype=torch.float64, device='cuda') for _ in range(8)]
        torch.cuda.synchronize()
        del junk                        # blocks go back to the allocator's free list, NaN-filled
        name = '%s/%s/nomask/it3' % (base, method)
        it, D, x = decomp.nmf.solve(y.copy(), D0.copy(), tol=0.0, minibatch=30, maxiter=3,
                                    method=method, random_seed=0)
        assert np.isfinite(D).all() and np.isfinite(x).all()
        assert np.max(np.abs(D - g[name + '/D'])) < 1e-8 * np.max(np.abs(g[name + '/D']))


def test_axpby_zero_coefficient_semantics():
    import torch
    from decomp_amd import nmf_minibatch
    D = torch.ones(4, 8, device='cuda')
    kern = nmf_minibatch._Kernels(D, 0)
    x = torch.full((4, 8), 2.0, device='cuda')
    y = torch.full((4, 8), float('nan'), device='cuda')
    kern.axpby(3.0, x, 0.0, y)                       # y never read
    assert torch.equal(y, torch.full_like(y, 6.0))
    z = torch.full((4, 8), float('inf'), device='cuda')
    z[0, 0] = 4.0
    kern.axpby(0.0, z, 0.5, z)                       # aliasing, a = 0: plain scaling, inf stays inf
    assert z[0, 0].item() == 2.0 and torch.isinf(z[1, 1]).item()


@pytest.mark.parametrize('lik', ['l2', 'kl'])
@pytest.mark.parametrize('masked', [False, True])
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_builtin_likelihood_methods_match_reference_parts(lik, masked, dtype):
    """Gaussian / Poisson .grad_x / .grad_d / .update_x / .update_d called directly, as the reference allows
    (grads.py:77-93, 108-125, 143-160): values AND shapes (Poisson without a mask returns the broadcastable
    [1, K] / [K, 1] sums) against the oracle's restatement of those lines."""
    import decomp_amd as decomp
    from oracle import nmf as onmf
    g = decomp.nmf_methods.grads
    rng = np.random.RandomState(11)
    y, D0, mask = _problem(rng, N=203, F=36, K=5, dtype=dtype)
    x = np.abs(rng.randn(203, 5)).astype(dtype)
    m = mask if masked else None
    obj = g.Gaussian() if lik == 'l2' else g.Poisson()
    tol = 1e-10 if dtype == np.float64 else 2e-5
    for mine, ref in ((obj.grad_x(y, x, D0, m), onmf._parts_x(y, x, D0, m, lik)),
                      (obj.grad_d(y, x, D0, m), onmf._parts_d(y, x, D0, m, lik))):
        for a, b in zip(mine, ref):
            assert isinstance(a, np.ndarray) and a.dtype == dtype and a.shape == b.shape, (a.shape, b.shape)
            assert np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b)))
    xu = obj.update_x(y, x, D0, m)
    du = obj.update_d(y, x, D0, m)
    assert np.max(np.abs(xu - onmf.update_x(y, x, D0, m, lik))) <= tol * max(1.0, np.max(xu))
    assert np.max(np.abs(du - onmf.update_d(y, x, D0, m, lik))) <= tol * max(1.0, np.max(du))


def test_gaussian_logp_and_poisson_logp_quirk():
    """Gaussian.logp (grads.py:127-135, incl. its `pi * 0.5` constant and the scale) with and without a
    (fractional) mask; Poisson.logp raises AttributeError as the reference's does (grads.py:162-165)."""
    import torch
    import decomp_amd as decomp
    g = decomp.nmf_methods.grads
    rng = np.random.RandomState(12)
    y, D0, mask = _problem(rng, N=150, F=33, K=4)
    x = np.abs(rng.randn(150, 4))
    frac = mask * rng.uniform(0.2, 1.0, size=mask.shape)
    for scale in (1.0, 2.5):
        for m in (None, mask, frac):
            loss = np.square((y - x.dot(D0)) / scale)
            want = -0.5 * loss - np.log(scale) - np.pi * 0.5
            want = np.sum(want) if m is None else np.sum(want * m)
            got = g.Gaussian(scale=scale).logp(y, x, D0, m)
            assert abs(got - want) <= 1e-10 * abs(want), (scale, got, want)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).cuda()
    got = g.Gaussian().logp(t(y), t(x), t(D0), None)
    assert got.is_cuda and got.dtype == torch.float32
    want = np.sum(-0.5 * np.square(y - x.dot(D0)) - np.pi * 0.5)
    assert abs(float(got) - want) <= 1e-5 * abs(want)
    with pytest.raises(AttributeError):
        g.Poisson().logp(y, x, D0, None)


def test_subclass_of_builtin_overriding_one_method():
    """The reference's extension point with inheritance (grads.py:96-160): `class Mine(Gaussian)` that
    overrides ONE gradient reaches the other through the base class (host loop), and a subclass that only
    overrides logp keeps the fused kernels -- both give the built-in 'l2' iterates."""
    import decomp_amd as decomp
    from oracle import nmf as onmf
    g = decomp.nmf_methods.grads
    rng = np.random.RandomState(13)
    y, D0, mask = _problem(rng)

    class OnlyGradX(g.Gaussian):
        calls = 0

        def grad_x(self, y, x, d, mask):
            OnlyGradX.calls += 1
            f = x @ d
            if mask is not None:
                f, y = f * mask, y * mask
            return y @ d.T, f @ d.T

    class OnlyLogp(g.Gaussian):
        def logp(self, y, x, d, mask):
            return 0.0

    assert g.fused_code(OnlyLogp()) == g.Gaussian._code and g.fused_code(OnlyGradX()) is None
    for m in (None, mask):
        ito, Do, xo = onmf.solve(y, D0.copy(), tol=1e-4, maxiter=40, mask=m)
        it, D, x = decomp.nmf.solve(y, D0.copy(), tol=1e-4, maxiter=40, likelihood=OnlyGradX(), mask=m)
        assert it == ito and np.max(np.abs(D - Do)) < 1e-9 and np.max(np.abs(x - xo)) < 1e-8 * max(1.0, np.max(xo))
        it2, D2, x2 = decomp.nmf.solve(y, D0.copy(), tol=1e-4, maxiter=40, likelihood=OnlyLogp(), mask=m)
        itb, Db, xb = decomp.nmf.solve(y, D0.copy(), tol=1e-4, maxiter=40, likelihood='l2', mask=m)
        assert it2 == itb and np.array_equal(D2, Db) and np.array_equal(x2, xb)
    assert OnlyGradX.calls > 0

    class KlOnlyGradD(g.Poisson):
        def grad_d(self, y, x, d, mask):
            f = x @ d + 1.0e-15
            if mask is None:
                return x.T @ (y / f), x.T.sum(axis=1, keepdims=True)
            return x.T @ ((y * mask) / f), x.T @ mask
    for m in (None, mask):
        ito, Do, xo = onmf.solve(y, D0.copy(), tol=0.0, maxiter=12, likelihood='kl', mask=m)
        it, D, x = decomp.nmf.solve(y, D0.copy(), tol=0.0, maxiter=12, likelihood=KlOnlyGradD(), mask=m)
        assert it == ito == 12 and np.max(np.abs(D - Do)) < 1e-9


@pytest.mark.parametrize('dtype', [np.float64, np.float32, np.complex128, np.complex64])
def test_linalg_inv_matches_numpy(dtype):
    """math_utils.linalg.inv (linalg.py:9-38): 2-D and batched (3-, 4-D) inputs against np.linalg.inv, incl. a
    matrix that needs row pivoting (zero leading entry) and the ADMM system A A^H + rho I."""
    import torch
    from decomp_amd.math_utils import linalg
    rng = np.random.RandomState(21)
    cplx = np.dtype(dtype).kind == 'c'

    def randn(*s):
        return ((rng.randn(*s) + 1j * rng.randn(*s)) if cplx else rng.randn(*s)).astype(dtype)
    tol = 2e-4 if dtype in (np.float32, np.complex64) else 1e-10
    for shape in [(7, 7), (3, 33, 33), (2, 3, 12, 12), (1, 1), (130, 130)]:
        x = randn(*shape)
        if shape == (7, 7):
            x[0, 0] = 0                                     # forces a row exchange in the first step
        got = linalg.inv(x)
        want = np.linalg.inv(x.astype(np.complex128 if cplx else np.float64))
        assert isinstance(got, np.ndarray) and got.dtype == dtype and got.shape == x.shape
        assert np.max(np.abs(got - want)) <= tol * max(1.0, np.max(np.abs(want))), shape
    A = randn(40, 90)
    S = A @ np.conj(A.T) + 1.0 * np.eye(40)
    got = linalg.inv(torch.from_numpy(S.astype(dtype)).cuda())
    assert got.is_cuda
    assert np.max(np.abs(got.cpu().numpy() @ S - np.eye(40))) < (5e-3 if tol > 1e-6 else 1e-9)
