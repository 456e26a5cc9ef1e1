"""GPU: edge cases of the three entry points -- degenerate iteration counts, single-row /
single-atom / single-channel problems, NaN propagation, sizes that are not multiples of any
tile -- checked against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    if np.isnan(b).any():      # a degenerate problem (0/0 in the reference too): same NaN pattern
        return np.array_equal(np.isnan(a), np.isnan(b)) and \
            np.allclose(a[~np.isnan(b)], b[~np.isnan(b)], rtol=tol, atol=tol)
    return float(np.max(np.abs(a - b))) <= tol * max(1.0, float(np.max(np.abs(b))))


@pytest.mark.parametrize('shape', [(1, 1, 1), (1, 7, 3), (5, 1, 2), (9, 4, 1), (131, 67, 5), (300, 129, 33)])
@pytest.mark.parametrize('lik', ['l2', 'kl'])
def test_nmf_odd_shapes(shape, lik):
    import decomp_amd
    from oracle import nmf as onmf
    N, F, K = shape
    rng = np.random.RandomState(N * 100 + F)
    y = np.abs(rng.randn(N, F)) + 0.05
    D0 = np.abs(rng.randn(K, F)) + 0.1
    mask = np.rint(rng.uniform(0.3, 1, size=(N, F)))
    for m in (None, mask):
        it, D, x = decomp_amd.nmf.solve(y, D0.copy(), tol=1e-9, maxiter=15, likelihood=lik, mask=m)
        ito, Do, xo = onmf.solve(y, D0.copy(), tol=1e-9, maxiter=15, likelihood=lik, mask=m)
        assert it == ito
        assert _close(D, Do, 1e-8) and _close(x, xo, 1e-8), (shape, lik, m is None)


@pytest.mark.parametrize('maxiter', [0, 1, 2])
def test_nmf_degenerate_maxiter(maxiter):
    """range(1, maxiter) runs no iteration for maxiter <= 1: (maxiter, l2_strict(D), x=ones)."""
    import decomp_amd
    from oracle import nmf as onmf
    rng = np.random.RandomState(0)
    y, D0 = np.abs(rng.randn(20, 6)), np.abs(rng.randn(3, 6)) + 0.1
    it, D, x = decomp_amd.nmf.solve(y, D0.copy(), tol=1e-3, maxiter=maxiter)
    ito, Do, xo = onmf.solve(y, D0.copy(), tol=1e-3, maxiter=maxiter)
    assert it == ito == maxiter
    assert _close(D, Do, 1e-12) and _close(x, xo, 1e-12)


def test_nmf_nan_never_converges():
    """A NaN in y poisons max|dD|; `nan < tol` is False, so the loop runs to maxiter like NumPy."""
    import decomp_amd
    rng = np.random.RandomState(1)
    y, D0 = np.abs(rng.randn(30, 8)), np.abs(rng.randn(2, 8)) + 0.1
    y[3, 4] = np.nan
    it, D, x = decomp_amd.nmf.solve(y, D0, tol=1e3, maxiter=6)
    assert it == 6 and np.isnan(D).any()


@pytest.mark.parametrize('method', ['ista', 'acc_ista', 'fista', 'cd'])
@pytest.mark.parametrize('maxiter', [1, 2, 11])
def test_lasso_small_iteration_counts(method, maxiter):
    from decomp_amd import lasso
    from oracle import lasso as olasso
    rng = np.random.RandomState(3)
    A, y = rng.randn(4, 9), rng.randn(6, 9)
    it, x = lasso.solve(y, A, 0.05, tol=1e-12, method=method, maxiter=maxiter)
    ito, xo = olasso.solve(y.copy(), A.copy(), 0.05, tol=1e-12, method=method, maxiter=maxiter)
    assert it == ito
    assert _close(x, xo, 1e-9), (method, maxiter)


def test_lasso_single_atom_and_wide():
    from decomp_amd import lasso
    from oracle import lasso as olasso
    rng = np.random.RandomState(4)
    for (Nb, F, K) in [(3, 5, 1), (1, 300, 7), (70, 3, 65), (2, 2, 130)]:
        A, y = rng.randn(K, F), rng.randn(Nb, F)
        for method in ('ista', 'cd', 'fista_pos'):
            it, x = lasso.solve(y, A, 0.01, tol=1e-10, method=method, maxiter=40)
            ito, xo = olasso.solve(y.copy(), A.copy(), 0.01, tol=1e-10, method=method, maxiter=40)
            assert it == ito and _close(x, xo, 1e-8), (Nb, F, K, method)


def test_lasso_given_x_is_not_mutated_and_used():
    from decomp_amd import lasso
    from oracle import lasso as olasso
    rng = np.random.RandomState(5)
    A, y, x0 = rng.randn(4, 9), rng.randn(6, 9), rng.randn(6, 4)
    keep = x0.copy()
    it, x = lasso.solve(y, A, 0.05, x=x0, tol=1e-12, method='acc_ista', maxiter=7)
    ito, xo = olasso.solve(y.copy(), A.copy(), 0.05, x=keep.copy(), tol=1e-12, method='acc_ista', maxiter=7)
    assert np.array_equal(x0, keep) and it == ito and _close(x, xo, 1e-9)


def test_dictionary_minibatch_equals_n_and_tail_rows():
    """minibatch == n_samples (one block per epoch) and a minibatch that leaves tail rows
    untouched that epoch (utils/data.py:115-121)."""
    from decomp_amd import dictionary_learning as dl
    from oracle import dictionary_learning as odl
    rng = np.random.RandomState(6)
    y, D0 = rng.randn(37, 6), rng.randn(4, 6)
    for mb in (37, 10):
        kw = dict(tol=0.0, minibatch=mb, maxiter=3, lasso_method='ista', lasso_iter=5, random_seed=2)
        it, D, x = dl.solve(y.copy(), D0.copy(), 0.05, **kw)
        ito, Do, xo = odl.solve(y.copy(), D0.copy(), 0.05, **kw)
        assert it == ito and _close(D, Do, 1e-8) and _close(x, xo, 1e-8), mb


@pytest.mark.parametrize('K', [1100, 2048])
def test_lasso_cd_many_atoms(K):
    """Coordinate descent with more than 1024 atoms (register-resident rows up to K = 2048):
    three sweeps against the oracle's as-written sweep."""
    from decomp_amd import lasso
    from oracle import lasso as olasso
    rng = np.random.RandomState(7)
    N, F = 6, 96
    A = rng.randn(K, F)
    xt = rng.randn(N, K) * (rng.uniform(size=(N, K)) < 0.01)
    y = xt @ A + 0.1 * rng.randn(N, F)
    it, x = lasso.solve(y, A, 0.05, tol=1e-12, method='cd', maxiter=3)
    ito, xo = olasso.solve(y.copy(), A.copy(), 0.05, tol=1e-12, method='cd', maxiter=3)
    assert it == ito == 2
    assert np.max(np.abs(x - xo)) <= 1e-9 * max(1.0, np.max(np.abs(xo)))
    assert np.count_nonzero(x) > 0


def test_stream_switch_orders_the_workspace_arena():
    """dcp_set_stream only records the new stream; the workspace arena shared by all calls of a handle is
    ordered lazily by the next arena-using call (ws_reserve -> ws_order_streams).  Interleave solves on two
    torch streams WITHOUT host synchronisation in between (each call returns after its own stream has the
    result; the next call on the other stream immediately reuses the arena): every result must equal the one
    computed alone."""
    import torch
    import decomp_amd
    rng = np.random.RandomState(4)
    probs = []
    for N, F, K in ((700, 260, 24), (1500, 512, 64), (300, 1000, 16)):
        xt = np.maximum(rng.randn(N, K), 0)
        Dt = np.maximum(rng.randn(K, F), 0)
        y = torch.from_numpy((xt @ Dt + 0.1 * np.abs(rng.randn(N, F))).astype(np.float32)).cuda()
        D0 = torch.from_numpy(np.maximum(Dt + 0.3 * rng.randn(K, F), 0.1).astype(np.float32)).cuda()
        probs.append((y, D0))
    alone = [decomp_amd.nmf.solve(y, D0, tol=0.0, maxiter=6) for y, D0 in probs]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(3):
        for i, (y, D0) in enumerate(probs):
            st = s1 if (i + rep) % 2 == 0 else s2
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append((i, decomp_amd.nmf.solve(y, D0, tol=0.0, maxiter=6)))
    torch.cuda.synchronize()
    for i, (it, D, x) in outs:
        assert it == alone[i][0]
        assert torch.equal(D, alone[i][1]) and torch.equal(x, alone[i][2]), i
