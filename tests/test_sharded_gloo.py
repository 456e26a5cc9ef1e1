"""CPU, world_size 2 over gloo: the host logic of the row-sharded MU loop
(decomp_amd.sharded.mu_loop: shard rows, ONE all-reduce of the [K, F+K] statistics per
iteration, replicated D update and stop test).  The per-rank arithmetic, which on a GPU
box is the HIP library, is supplied here by the CPU oracle acting as a stand-in backend --
this tests the partitioning/collective logic, not the kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleStepBackend(object):
    """Same interface as decomp_amd.sharded.HipStepBackend, arithmetic by the oracle."""

    def __init__(self, y, mask, x, lik):
        from oracle import nmf as onmf
        self.o = onmf
        self.y, self.mask, self.x, self.lik = y, mask, x, lik
        self.md = [0.0, 0.0]

    def local_stats(self, D):
        Dn = D.numpy()
        self._x_prev = self.x
        self.x = self.o.update_x(self.y, self.x, Dn, self.mask, self.lik)
        pos, neg = self.o._parts_d(self.y, self.x, Dn, self.mask, self.lik)
        if self.lik == 'l2' and self.mask is None:
            stats = np.concatenate([self.x.T @ self.y, self.x.T @ self.x], axis=1)
        else:
            stats = np.concatenate([pos, np.broadcast_to(neg, pos.shape)], axis=1)
        return torch.from_numpy(np.ascontiguousarray(stats))

    def rollback(self):
        self.x = self._x_prev

    def update(self, stats, D, D_new, slot):
        from oracle.common import l2_strict
        s, Dn = stats.numpy(), D.numpy()
        F = Dn.shape[1]
        if self.lik == 'l2' and self.mask is None:
            num, den = s[:, :F], s[:, F:] @ Dn
        else:
            num, den = s[:, :F], s[:, F:]
        out = l2_strict(Dn * np.maximum(num, 0) / np.maximum(den, 1e-15))
        self.md[slot] = float(np.max(np.abs(Dn - out)))
        D_new.copy_(torch.from_numpy(out))

    def read_maxdiff(self, slot):
        return self.md[slot]


def _problem(lik, masked):
    rng = np.random.RandomState(5)
    N, F, K = 96, 20, 3
    xt, Dt = np.maximum(rng.randn(N, K), 0), np.maximum(rng.randn(K, F), 0)
    y = xt @ Dt + 0.1 * np.abs(rng.randn(N, F))
    D0 = np.maximum(Dt + 0.3 * rng.randn(K, F), 0.1)
    mask = np.rint(rng.uniform(0.3, 1, size=(N, F))) if masked else None
    return y, D0, mask


def _worker(rank, world, port, lik, masked, tol, maxiter, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from decomp_amd import sharded
        from oracle.common import l2_strict
        y, D0, mask = _problem(lik, masked)
        rows = slice(rank * len(y) // world, (rank + 1) * len(y) // world)
        x = np.ones((rows.stop - rows.start, D0.shape[0]))
        be = OracleStepBackend(y[rows], None if mask is None else mask[rows], x, lik)
        D = torch.from_numpy(l2_strict(D0))
        it, Dout = sharded.mu_loop(be, D, tol, maxiter, world_size=world,
                                   new_like=torch.empty_like)
        q.put((rank, it, Dout.numpy().copy(), be.x.copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('lik,masked', [('l2', False), ('l2', True), ('kl', False), ('kl', True)])
def test_two_rank_sharded_loop_equals_single_process(lik, masked):
    from oracle import nmf as onmf
    world, tol, maxiter = 2, 1e-5, 120
    y, D0, mask = _problem(lik, masked)
    it_ref, D_ref, x_ref = onmf.solve(y, D0.copy(), tol=tol, maxiter=maxiter, likelihood=lik,
                                      mask=mask)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, lik, masked, tol, maxiter, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    its = [r[1] for r in res]
    assert its[0] == its[1] == it_ref                 # identical decision on every rank
    assert np.array_equal(res[0][2], res[1][2])        # replicated D is bit-identical
    assert np.allclose(res[0][2], D_ref, rtol=1e-9, atol=1e-12)
    x_all = np.concatenate([r[3] for r in res], axis=0)
    assert np.allclose(x_all, x_ref, rtol=1e-8, atol=1e-12)


# ---------------------------------------------------------------- dictionary learning -----
class OracleDictBackend(object):
    """Same interface as decomp_amd.sharded.HipDictBackend, arithmetic by the oracle."""

    def __init__(self, lasso_method, lasso_iter, lasso_tol, alpha, K, F, dtype=torch.float64):
        self.lm, self.li, self.lt, self.alpha = lasso_method, lasso_iter, lasso_tol, alpha
        self.stats = torch.zeros((K, F + K), dtype=dtype)
        self._md = 0.0

    def local_stats(self, y_rows, x_rows, D):
        from oracle import lasso as olasso
        yn, xn, Dn = y_rows.numpy(), x_rows.numpy(), D.numpy()
        _, xs = olasso.solve_fastpath(yn, Dn, self.alpha, xn, self.lt, self.li, self.lm)
        xn[...] = xs
        xH = np.conj(xn.T)
        self.stats.copy_(torch.from_numpy(np.ascontiguousarray(np.concatenate([xH @ yn, xH @ xn], axis=1))))
        return self.stats

    def update_async(self, stats, beta, A, B, D, D_new):
        from oracle.dictionary_learning import atom_sweep
        s = stats.numpy()
        F = D.shape[1]
        A[...] = torch.from_numpy(beta * A.numpy() + s[:, F:])
        B[...] = torch.from_numpy(beta * B.numpy() + s[:, :F])
        out = atom_sweep(D.numpy(), A.numpy(), B.numpy())
        D_new.copy_(torch.from_numpy(out))
        self._md = float(np.max(np.abs(D.numpy() - out)))

    def maxdiff_token(self, slot):
        return self._md

    def read_maxdiff(self, token):
        return token

    def gather(self, src, index, n, out):
        if n:
            out[:n] = src[index]

    def scatter(self, src, index, n, out):
        if n:
            out[index] = src[:n]


def _dl_problem(cplx=False):
    rng = np.random.RandomState(11)
    if cplx:        # configs[4]'s form: complex64 data and statistics
        def rn(*sh):
            return rng.randn(*sh) + 1j * rng.randn(*sh)
        Dt = rn(3, 5)
        xt = rn(103, 3) * rng.uniform(size=(103, 3))
        y = xt @ Dt + 0.1 * rn(103, 5)
        return y.astype(np.complex64), (Dt + 0.2 * rn(3, 5)).astype(np.complex64)
    Dt = rng.randn(3, 5)
    xt = rng.randn(103, 3) * rng.uniform(size=(103, 3))
    y = xt @ Dt + 0.1 * rng.randn(103, 5)
    return y, Dt + 0.2 * rng.randn(3, 5)


def _dl_worker(rank, world, port, q, tol, cplx=False):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from decomp_amd import sharded
        from oracle.common import l2_strict
        y, D0 = _dl_problem(cplx)
        tdt = torch.complex64 if cplx else torch.float64
        N = y.shape[0]
        bounds = [0, 40, N]                       # unequal shards: 40 and 63 rows
        lo, hi = bounds[rank], bounds[rank + 1]
        be = OracleDictBackend('ista', 8, 1e-5, 0.1, 3, 5, dtype=tdt)
        D = torch.from_numpy(l2_strict(D0))
        assert D.dtype == tdt
        rng = np.random.RandomState(4)
        calls = {'n': 0}
        real_all_reduce = dist.all_reduce

        def counting_all_reduce(*a, **k):
            calls['n'] += 1
            return real_all_reduce(*a, **k)
        dist.all_reduce = counting_all_reduce
        it, Dout, xout = sharded.dict_loop(
            be, torch.from_numpy(y[lo:hi].copy()), torch.ones((hi - lo, 3), dtype=tdt), lo, N,
            D, tol, 25, 4, rng, torch.empty_like, lambda shape: torch.zeros(shape, dtype=tdt),
            lambda idx: torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)), world_size=world)
        q.put((rank, it, Dout.numpy().copy(), xout.numpy().copy(), calls['n']))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('tol,cplx', [(0.0, False), (0.05, False), (0.0, True), (0.05, True)])
def test_two_rank_sharded_dictionary_learning_equals_single_process(tol, cplx):
    """Rows owned by the ranks for the whole run (40 + 63 of 103), the global minibatch composition of
    the shared RandomState, ONE all-reduce per minibatch step and nothing else; with tol > 0 the lagged
    stop test must return exactly the reference's iteration, dictionary and codes.  cplx: complex64 data,
    the multi-GPU form of configs[4] -- the complex [K, F+K] statistics x^H [y | x]
    (dictionary_learning.py:147-152) cross the all-reduce."""
    from oracle import dictionary_learning as odl
    y, D0 = _dl_problem(cplx)
    it_ref, D_ref, x_ref = odl.solve(y.copy(), D0.copy(), 0.1, tol=tol, minibatch=25, maxiter=4,
                                     lasso_method='ista', lasso_iter=8, lasso_tol=1e-5, random_seed=4)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_dl_worker, args=(r, 2, port, q, tol, cplx)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == it_ref
    if tol > 0:
        assert it_ref < 4                        # the stop test really fired
    assert np.array_equal(res[0][2], res[1][2])                  # replicated D is bit-identical
    rt, at = (2e-4, 2e-5) if cplx else (1e-9, 1e-12)             # single precision: summation order of the statistics
    assert res[0][2].dtype == D_ref.dtype == (np.complex64 if cplx else np.float64)
    assert np.allclose(res[0][2], D_ref, rtol=rt, atol=at)
    x_all = np.concatenate([res[0][3], res[1][3]], axis=0)       # rank order = original row order
    assert np.allclose(x_all, x_ref, rtol=10 * rt, atol=10 * at)
    # exactly one collective per executed minibatch step (the speculative step after a passed stop
    # test included), the same number on both ranks
    assert res[0][4] == res[1][4]
    n_loop = 103 // 25
    assert res[0][4] <= (4 - 1) * n_loop and res[0][4] >= 1
    if tol == 0.0:
        assert res[0][4] == (4 - 1) * n_loop
