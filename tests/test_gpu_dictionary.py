"""GPU parity: decomp_amd.dictionary_learning (HIP, through the C ABI) against the golden
vectors of the real reference (tests/test_dictionary.py shapes, real and complex, every
starred lasso method, 1-3 epochs with the exact RandomState shuffle) and the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _g():
    return np.load(os.path.join(GOLDEN, 'dl_golden.npz'), allow_pickle=False)


def _err(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(1.0, float(np.max(np.abs(b))))


def _cases():
    g = _g()
    return [str(c) for c in g['cases']]


@pytest.mark.parametrize('name', _cases())
def test_golden(name):
    from decomp_amd import dictionary_learning as dl
    g = _g()
    parts = name.split('/')
    base = parts[0]
    y, D0 = g[base + '/y'], g[base + '/D0']
    if parts[1] == 'reftest':
        it, D, x = dl.solve(y.copy(), D0.copy(), 0.1, tol=1.0e-4, minibatch=100, maxiter=1000,
                            lasso_method='acc_ista', lasso_iter=1000, random_seed=0)
    else:
        minibatch = int(parts[1][2:])
        lm = parts[2].rstrip('0123456789')
        li = int(parts[2][len(lm):])
        epochs = int(parts[4][2:])
        use_mask = parts[3] == 'mask'
        mask = g[base + '/mask']
        yy = y * mask if use_mask else y
        it, D, x = dl.solve(yy.copy(), D0.copy(), 0.1, tol=0.0, minibatch=minibatch,
                            maxiter=epochs + 1, lasso_method=lm, lasso_iter=li,
                            lasso_tol=1.0e-5, random_seed=0,
                            mask=mask.copy() if use_mask else None)
    assert it == int(g[name + '/it']), name
    assert D.dtype == y.dtype and x.shape == (101, 3)
    assert _err(D, g[name + '/D']) < 1e-7, (name, _err(D, g[name + '/D']))
    assert _err(x, g[name + '/x']) < 1e-7, (name, _err(x, g[name + '/x']))


def _gx():
    return np.load(os.path.join(GOLDEN, 'dl_extra_golden.npz'), allow_pickle=False)


def _extra_cases():
    return [str(c) for c in _gx()['cases']]


@pytest.mark.parametrize('name', _extra_cases())
def test_golden_single_precision_and_wide(name):
    """Reference-generated fixtures (oracle/make_golden.py::gen_dl_extra): float32 / complex64 runs of
    the reference's own test shapes (the dtypes of BASELINE configs[2] and [4]) and wide dictionaries,
    K = 160 real (two full 64-atom blocks + a 32-atom tail of the blocked atom sweep) and K = 80
    complex, in all four dtypes."""
    from decomp_amd import dictionary_learning as dl
    from test_oracle_golden import _run_dl_case
    g = _gx()
    it, D, x = _run_dl_case(dl.solve, g, name)
    Dg, xg = g[name + '/D'], g[name + '/x']
    assert it == int(g[name + '/it']), name
    assert D.dtype == Dg.dtype and x.dtype == xg.dtype and x.shape == xg.shape
    single = Dg.dtype in (np.float32, np.complex64)
    # double precision: rounding-level agreement (the blocked sweep forms norms through a Gram
    # matrix); single precision: 2e-4 of the largest entry, the bound the CPU oracle itself meets
    tol_D = 2e-4 if single else 1e-7
    tol_x = 5e-4 if single else 1e-7
    assert _err(D, Dg) < tol_D, (name, _err(D, Dg))
    assert _err(x, xg) < tol_x, (name, _err(x, xg))


@pytest.mark.parametrize('dt,K,F', [('float64', 160, 96), ('float32', 160, 96), ('float64', 200, 96),
                                    ('complex128', 80, 96), ('complex64', 80, 96), ('float64', 64, 96),
                                    ('float32', 33, 96),
                                    # float32 with K, F multiples of 64: the fused three-launch path
                                    # (csrc/atom_fused_f32.hpp), 1 / 3 / 4 blocks
                                    ('float32', 64, 128), ('float32', 192, 128), ('float32', 256, 320),
                                    # ragged channel counts (bounds-checked products), complex64 on the
                                    # planar-rows products with 2, 3 and 4 blocks of 32 atoms
                                    ('float32', 128, 100), ('complex64', 96, 130), ('complex64', 64, 256),
                                    ('complex64', 128, 67), ('float64', 96, 130),
                                    # the widths the dictionary step is TIMED at (BASELINE configs[2] / [4]):
                                    # 8 fused float32 blocks over F = 4096, 16 complex blocks of 32 atoms
                                    ('float32', 512, 4096), ('complex64', 512, 8192), ('float64', 512, 1024),
                                    ('complex128', 512, 512)])
def test_atom_sweep_direct_against_oracle(dt, K, F):
    """dcp_dict_update_* alone (A/B accumulation + the blocked Gauss-Seidel atom sweep + max|dD|)
    against oracle.dictionary_learning.atom_sweep (dictionary_learning.py:154-159) for dictionaries
    wider than one 64-atom block, incl. a tail block and a zero code column (the A_kk + 1e-15
    blow-up that l2 then renormalises)."""
    import ctypes
    import torch
    from decomp_amd import _arrays, _hip
    from oracle import dictionary_learning as odl
    from oracle.common import l2_strict
    rng = np.random.RandomState(K)
    cplx = dt.startswith('complex')
    Nb = 300 if K <= 256 else 1200

    def randn(*s):
        return (rng.randn(*s) + 1j * rng.randn(*s)) if cplx else rng.randn(*s)
    x = (randn(Nb, K) * (rng.uniform(size=(Nb, K)) < 0.15)).astype(dt)
    x[:, 5] = 0                                           # an unused atom
    y = randn(Nb, F).astype(dt)
    D = l2_strict(randn(K, F)).astype(dt)
    A_old = (np.conj(x.T) @ x).astype(dt) * 0.5
    B_old = (np.conj(x.T) @ y).astype(dt) * 0.5
    beta = 0.75
    stats = np.concatenate([np.conj(x.T) @ y, np.conj(x.T) @ x], axis=1).astype(dt)
    A_ref = beta * A_old + stats[:, F:]
    B_ref = beta * B_old + stats[:, :F]
    D_ref = odl.atom_sweep(D, A_ref, B_ref)

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    st, A, B, Dd = t(stats), t(A_old), t(B_old), t(D)
    D_new = torch.empty_like(Dd)
    rdt = torch.float32 if dt in ('float32', 'complex64') else torch.float64
    md = torch.zeros((1,), dtype=rdt, device='cuda')
    lib, h = _arrays.lib_handle(Dd)
    fn = getattr(lib, 'dcp_dict_update_' + _arrays.suffix(Dd))
    _hip.check(h, fn(h, _arrays.ptr(st), beta, _arrays.ptr(A), _arrays.ptr(B), _arrays.ptr(Dd),
                     _arrays.ptr(D_new), F, K, _arrays.ptr(md)), 'dcp_dict_update')
    torch.cuda.synchronize()
    single = dt in ('float32', 'complex64')
    tol = 2e-4 if single else 1e-9
    assert _err(A.cpu().numpy(), A_ref) < (1e-5 if single else 1e-12)
    assert _err(B.cpu().numpy(), B_ref) < (1e-5 if single else 1e-12)
    assert _err(D_new.cpu().numpy(), D_ref) < tol, _err(D_new.cpu().numpy(), D_ref)
    want_md = float(np.max(np.abs(D - D_ref)))
    assert abs(float(md.item()) - want_md) <= tol * max(1.0, want_md)


@pytest.mark.parametrize('dt', ['float32', 'complex64'])
@pytest.mark.parametrize('lm', ['ista', 'cd', 'admm', 'ista_pos', 'cd_pos', 'parallel_cd'])
def test_against_oracle_medium(dt, lm):
    """1024 x 256, K = 64, minibatch 256, 2 epochs: float32 / complex64 (MFMA path for
    float32) against the CPU oracle."""
    from decomp_amd import dictionary_learning as dl
    from oracle import dictionary_learning as odl
    rng = np.random.RandomState(2)
    N, F, K = 1024, 256, 64
    cplx = dt == 'complex64'
    if cplx and lm.endswith('_pos'):
        pytest.skip('positive solvers are real only')

    def randn(*s):
        return (rng.randn(*s) + 1j * rng.randn(*s)) if cplx else rng.randn(*s)
    Dt = randn(K, F)
    xt = 3.0 * randn(N, K) * (rng.uniform(size=(N, K)) < 0.1)
    y = (xt @ Dt + 0.1 * randn(N, F)).astype(dt)
    D0 = (Dt + 0.2 * randn(K, F)).astype(dt)
    kw = dict(tol=0.0, minibatch=256, maxiter=3, lasso_method=lm, lasso_iter=10,
              lasso_tol=1e-5, random_seed=0)
    it, D, x = dl.solve(y.copy(), D0.copy(), 0.01, **kw)
    ito, Do, xo = odl.solve(y.copy(), D0.copy(), 0.01, **kw)
    assert it == ito == 3
    assert _err(D, Do) < 5e-4, _err(D, Do)
    assert _err(x, xo) < 5e-3, _err(x, xo)
    assert np.count_nonzero(x) > 0


def test_reference_property_minimum():
    """tests/test_dictionary.py:45-56: converges, and the loss at the solution is below the
    loss at random perturbations of it."""
    from decomp_amd import dictionary_learning as dl
    from oracle.common import l2
    g = _g()
    y, D0 = g['dl_f64/y'], g['dl_f64/D0']
    alpha = 0.1
    it, D, x = dl.solve(y, D0.copy(), alpha, x=None, tol=1.0e-4, minibatch=100, maxiter=1000,
                        lasso_method='acc_ista', lasso_iter=1000, random_seed=0)
    assert it < 999 and not np.allclose(x, 0)

    def loss(x, D):
        a = alpha * y.shape[1]
        return np.sum(0.5 / a * np.abs(y - x @ l2(D)) ** 2) + np.sum(np.abs(x))
    rng = np.random.RandomState(0)
    base = loss(x, D)
    for _ in range(3):
        assert base < loss(x + rng.randn(*x.shape) * 1e-3, D + rng.randn(*D.shape) * 1e-3)


class _RefDictProblem(object):
    """setUp of tests/test_dictionary.py:35-43 (fresh per test method, as unittest does)."""
    alpha = 0.1

    def __init__(self, cplx):
        self.rng = np.random.RandomState(0)
        self.cplx = cplx
        self.Dtrue = self.randn(3, 5)
        self.xtrue = self.randn(101, 3) * self.rng.uniform(size=303).reshape(101, 3)
        self.y = self.xtrue @ self.Dtrue + self.randn(101, 5) * 0.1
        self.D = self.Dtrue + self.randn(3, 5) * 0.2
        self.mask = np.rint(self.rng.uniform(0.45, 1, size=505)).reshape(101, 5)

    def randn(self, *s):
        return (self.rng.randn(*s) + self.rng.randn(*s) * 1.0j) if self.cplx else self.rng.randn(*s)

    def error(self, x, D, m=None):
        from oracle.common import l2
        m = np.ones(self.y.shape) if m is None else m
        a = self.alpha * np.sum(m, axis=-1, keepdims=True)
        return np.sum(0.5 / a * np.square(np.abs(self.y - x @ l2(D))) * m) + np.sum(np.abs(x))

    def assert_minimum(self, x, D, tol, n, m=None):
        base = self.error(x, D, m)
        for _ in range(n):
            dx = self.randn(*x.shape) * tol
            dD = self.randn(*D.shape) * tol
            assert base < self.error(x + dx, D + dD, m)


@pytest.mark.parametrize('cplx', [False, True])
def test_reference_property_run(cplx):
    """tests/test_dictionary.py:45-56 (TestFloat / TestComplex.test_run)."""
    from decomp_amd import dictionary_learning as dl
    p = _RefDictProblem(cplx)
    it, D, x = dl.solve(p.y, p.D.copy(), p.alpha, x=None, tol=1.0e-4, method='block_cd', minibatch=100,
                        maxiter=1000, lasso_method='acc_ista', lasso_iter=1000, random_seed=0)
    assert it < 1000 - 1
    p.assert_minimum(x, D, tol=1.0e-3, n=3)
    assert not np.allclose(x, 0.0)


@pytest.mark.parametrize('cplx', [False, True])
def test_reference_property_run_mask(cplx):
    """tests/test_dictionary.py:59-81 (test_run_mask; the reference checks the minimum against the
    INITIAL dictionary, :74-75, and that a different minibatch size ends elsewhere)."""
    from decomp_amd import dictionary_learning as dl
    p = _RefDictProblem(cplx)
    it, D, x = dl.solve(p.mask * p.y, p.D.copy(), p.alpha, x=None, tol=1.0e-4, minibatch=100,
                        maxiter=1000, lasso_method='acc_ista', lasso_iter=1000, random_seed=0,
                        mask=p.mask)
    assert it < 1000 - 1
    assert not np.allclose(x, 0.0)
    Dinit = p.D.copy()
    p.assert_minimum(x, Dinit, tol=1.0e-3, n=3, m=p.mask)
    it2, D2, x2 = dl.solve(p.y, Dinit, p.alpha, x=None, tol=1.0e-5, minibatch=10, maxiter=1000,
                           lasso_method='acc_ista', lasso_iter=1000, random_seed=0)
    assert not np.allclose(Dinit, D2, atol=1.0e-4)


def test_minibatch_container_roundtrip():
    """tests/test_utils.py: shuffle is cumulative, .array restores the original order, tail
    rows are skipped by the iteration."""
    import torch
    from decomp_amd.utils.data import MinibatchData
    a = np.arange(23 * 3, dtype=np.float64).reshape(23, 3)
    mb = MinibatchData(torch.from_numpy(a).cuda(), 5)
    assert mb.n_loop == 4
    rng = np.random.RandomState(0)
    idx = np.arange(23)
    ref = a.copy()
    for _ in range(3):
        rng.shuffle(idx)
        mb.shuffle(idx)
        ref = ref[idx]
        blocks = [b.cpu().numpy() for b in mb]
        assert len(blocks) == 4 and np.array_equal(np.concatenate(blocks), ref[:20])
        assert np.array_equal(mb.array.cpu().numpy(), a)
    with pytest.raises(ValueError):
        MinibatchData(torch.zeros((3, 2)).cuda(), 5)


def test_errors():
    from decomp_amd import dictionary_learning as dl
    y, D = np.random.randn(20, 5), np.random.randn(3, 5)
    with pytest.raises(NotImplementedError):
        dl.solve(y, D, 0.1)                                  # minibatch is required
    with pytest.raises(NotImplementedError):
        dl.solve(y, D, 0.1, minibatch=10, method='parallel_cd')
    with pytest.raises(ValueError):
        dl.solve(y, D, 0.1, minibatch=50)                    # minibatch > n_samples


def _sharded_dl_problem(cplx=False):
    rng = np.random.RandomState(21)
    N, F, K = 240, 64, 12
    if cplx:        # configs[4]'s form: complex64 data, complex [K, F+K] statistics through the all-reduce
        def rn(*sh):
            return rng.randn(*sh) + 1j * rng.randn(*sh)
        Dt = rn(K, F)
        xt = 3.0 * rn(N, K) * (rng.uniform(size=(N, K)) < 0.2)
        y = (xt @ Dt + 0.1 * rn(N, F)).astype(np.complex64)
        return y, (Dt + 0.2 * rn(K, F)).astype(np.complex64)
    Dt = rng.randn(K, F)
    xt = 3.0 * rng.randn(N, K) * (rng.uniform(size=(N, K)) < 0.2)
    y = (xt @ Dt + 0.1 * rng.randn(N, F)).astype(np.float32)
    return y, (Dt + 0.2 * rng.randn(K, F)).astype(np.float32)


_DL_KW = dict(tol=0.0, minibatch=48, maxiter=3, lasso_method='ista', lasso_iter=6, lasso_tol=1e-5,
              random_seed=3)


def test_sharded_dictionary_world1_equals_solve():
    from decomp_amd import dictionary_learning as dl, sharded
    y, D0 = _sharded_dl_problem()
    a = dl.solve(y.copy(), D0.copy(), 0.02, **_DL_KW)
    b = sharded.dictionary_learning_sharded(y.copy(), D0.copy(), 0.02, **_DL_KW)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def _dl_gloo_gpu_worker(rank, world, port, q, kw=None, cplx=False):
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from decomp_amd import sharded
        y, D0 = _sharded_dl_problem(cplx)
        lo, hi = (0, 100) if rank == 0 else (100, y.shape[0])      # unequal shards, owned for the whole run
        calls = {'n': 0}
        real = dist.all_reduce

        def counting(*a, **k):
            calls['n'] += 1
            return real(*a, **k)
        dist.all_reduce = counting
        it, D, x = sharded.dictionary_learning_sharded(y[lo:hi].copy(), D0.copy(), 0.02, **(kw or _DL_KW))
        q.put((rank, it, D, x, calls['n']))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('cplx', [False, True])
def test_sharded_dictionary_two_ranks_on_one_gpu_gloo(cplx):
    """Samples sharded over two processes sharing the test box's GPU (each owns its rows of y and x for
    the whole run); ONLY the [K, F+K] statistics cross ranks, one all-reduce per minibatch step over
    gloo: must reproduce the single-process result to rounding.  cplx: complex64 data -- the multi-GPU form
    of configs[4]; the complex x^H [y | x] statistics cross the all-reduce (dictionary_learning.py:147-152)."""
    import os
    import torch.multiprocessing as mp
    from decomp_amd import dictionary_learning as dl
    y, D0 = _sharded_dl_problem(cplx)
    it1, D1, x1 = dl.solve(y.copy(), D0.copy(), 0.02, **_DL_KW)
    assert D1.dtype == (np.complex64 if cplx else np.float32)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 1000) + (1000 if cplx else 0)
    procs = [ctx.Process(target=_dl_gloo_gpu_worker, args=(r, 2, port, q, None, cplx)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == it1
    assert np.array_equal(res[0][2], res[1][2])                       # replicated D: bit-identical
    x_all = np.concatenate([res[0][3], res[1][3]], axis=0)            # rank order = original row order
    assert res[0][3].shape[0] == 100 and x_all.shape == x1.shape
    assert _err(res[0][2], D1) < 1e-4 and _err(x_all, x1) < 1e-3
    n_steps = (_DL_KW['maxiter'] - 1) * (y.shape[0] // _DL_KW['minibatch'])
    assert res[0][4] == res[1][4] == n_steps                          # exactly one collective per step


def test_sharded_dictionary_early_exit_per_rank():
    """with a LOOSE lasso_tol the LASSO's early exit (lasso.py:293, checked on iterations 0, 10, ...)
    fires, and in the sharded run it is taken per rank on the rows that rank owns.  The two-rank result must
    stay within O(lasso_tol) of the single-process one (codes) and the replicated D must still be bit-identical
    across ranks; the single-process run with the same settings really does stop early (its codes differ from
    a run that is forced through all iterations)."""
    import os
    import torch.multiprocessing as mp
    from decomp_amd import dictionary_learning as dl
    y, D0 = _sharded_dl_problem()
    lasso_tol = 0.05
    kw = dict(_DL_KW, lasso_iter=41, lasso_tol=lasso_tol)
    it1, D1, x1 = dl.solve(y.copy(), D0.copy(), 0.02, **kw)
    itf, Df, xf = dl.solve(y.copy(), D0.copy(), 0.02, **dict(kw, lasso_tol=0.0))
    assert np.max(np.abs(x1 - xf)) > 0                                  # the early exit was taken
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 28800 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_dl_gloo_gpu_worker, args=(r, 2, port, q, kw)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == it1
    assert np.array_equal(res[0][2], res[1][2])                         # replicated D: bit-identical
    x_all = np.concatenate([res[0][3], res[1][3]], axis=0)
    assert np.max(np.abs(x_all - x1)) <= 20 * lasso_tol                 # O(lasso_tol), codes are O(3)
    assert _err(res[0][2], D1) < 0.1


def test_stop_test_fires_mid_epoch_speculative_step_is_discarded():
    """Round 4: dictionary_learning.solve reads max|D - D_new| < tol (dictionary_learning.py:161-162) one step late and
    drops the step it had enqueued speculatively (its codes are not scattered, its D_new is not returned).  A run whose
    stop test fires in the MIDDLE of an epoch must return exactly what the oracle returns: same epoch number, the
    dictionary of the converged step, and codes in the original row order with only the rows visited so far updated."""
    from decomp_amd import dictionary_learning as dl
    from oracle import dictionary_learning as odl
    rng = np.random.RandomState(17)
    N, F, K, mb = 203, 24, 5, 20
    Dt = rng.randn(K, F)
    xt = 2.0 * rng.randn(N, K) * (rng.uniform(size=(N, K)) < 0.4)
    y = xt @ Dt + 0.05 * rng.randn(N, F)
    D0 = Dt + 0.2 * rng.randn(K, F)
    trace = []
    odl.solve(y.copy(), D0.copy(), 0.02, tol=0.0, minibatch=mb, maxiter=4, lasso_method='ista', lasso_iter=12,
              lasso_tol=1e-7, random_seed=5, trace=trace)
    n_loop = N // mb
    # a tolerance between the max|dD| of two consecutive steps in the middle of epoch 2
    s = n_loop + 4
    diffs = [t['maxdiff'] for t in trace]
    first_below = None
    for tol_step in range(s, len(diffs)):
        cand = 0.5 * (diffs[tol_step] + min(diffs[:tol_step]))
        if diffs[tol_step] < cand and all(d >= cand for d in diffs[:tol_step]) and tol_step % n_loop not in (0, n_loop - 1):
            first_below = (tol_step, cand)
            break
    assert first_below is not None, diffs
    step, tol = first_below
    kw = dict(tol=tol, minibatch=mb, maxiter=4, lasso_method='ista', lasso_iter=12, lasso_tol=1e-7, random_seed=5)
    ito, Do, xo = odl.solve(y.copy(), D0.copy(), 0.02, **kw)
    it, D, x = dl.solve(y.copy(), D0.copy(), 0.02, **kw)
    assert ito == step // n_loop + 1 and it == ito
    assert _err(D, Do) < 1e-8 and _err(x, xo) < 1e-8
    assert np.any(x == 1.0)            # rows not yet visited in the interrupted epoch (and the skipped tail) keep x = 1


def test_async_step_and_registered_prefetch_equal_the_synchronous_step():
    """dcp_dict_step_async_* (max|dD| left in caller memory, no host wait) with a row gather registered through
    dcp_dict_prefetch_rows_bytes must leave x, D_new, A, B and max|dD| bit-identical to dcp_dict_step_*, and the
    prefetched block must equal the plain gather."""
    import ctypes
    import torch
    from decomp_amd import _arrays, _hip
    g = torch.Generator(device='cuda')
    g.manual_seed(3)
    for dt, cplx in ((torch.float32, False), (torch.complex64, True)):
        MB, F, K, N = 512, 256, 64, 2048

        def randn(*sh):
            r = torch.randn(sh, generator=g, device='cuda')
            return torch.complex(r, torch.randn(sh, generator=g, device='cuda')) if cplx else r
        Yall = randn(N, F)
        D = randn(K, F)
        _arrays.l2_normalize_(D, strict=True)
        idx = torch.randperm(N, generator=g, device='cuda')[:MB].to(torch.int64)
        idx_next = torch.randperm(N, generator=g, device='cuda')[:MB].to(torch.int64)
        Y = Yall[idx].contiguous()
        sfx = 'c64' if cplx else 'f32'
        lib, h = _arrays.lib_handle(D)
        outs = []
        for mode in ('sync', 'async'):
            x = torch.ones((MB, K), device='cuda', dtype=dt)
            A = torch.zeros((K, K), device='cuda', dtype=dt)
            B = torch.zeros((K, F), device='cuda', dtype=dt)
            Dn = torch.empty_like(D)
            lit = ctypes.c_int(0)
            if mode == 'sync':
                md = ctypes.c_double(0)
                _hip.check(h, getattr(lib, 'dcp_dict_step_' + sfx)(
                    h, _arrays.ptr(Y), _arrays.ptr(x), _arrays.ptr(D), _arrays.ptr(Dn), _arrays.ptr(A), _arrays.ptr(B),
                    MB, F, K, (1.0 - MB) / 1.0, 0.02, _hip.LASSO_ISTA, 10, 1e-6, ctypes.byref(md), ctypes.byref(lit)), 'step')
                mdv = md.value
                staged = None
            else:
                md = torch.zeros((1,), device='cuda', dtype=torch.float32)
                staged = torch.zeros((MB, F), device='cuda', dtype=dt)
                _hip.check(h, lib.dcp_dict_prefetch_rows_bytes(h, _arrays.ptr(Yall), _arrays.ptr(idx_next), MB,
                                                               F * Yall.element_size(), _arrays.ptr(staged)), 'prefetch')
                _hip.check(h, getattr(lib, 'dcp_dict_step_async_' + sfx)(
                    h, _arrays.ptr(Y), _arrays.ptr(x), _arrays.ptr(D), _arrays.ptr(Dn), _arrays.ptr(A), _arrays.ptr(B),
                    MB, F, K, (1.0 - MB) / 1.0, 0.02, _hip.LASSO_ISTA, 10, 1e-6, _arrays.ptr(md), ctypes.byref(lit)), 'step_async')
                torch.cuda.synchronize()
                mdv = float(md.item())
            outs.append((x, Dn, A, B, mdv, lit.value, staged))
        a, b = outs
        for i in range(4):
            assert torch.equal(a[i], b[i]), (sfx, i)
        assert a[5] == b[5] and abs(a[4] - b[4]) <= 1e-12 * max(1.0, abs(a[4]))
        assert torch.equal(b[6], Yall[idx_next])


@pytest.mark.gpu
@pytest.mark.parametrize('dt', ['float32', 'float64', 'complex64'])
def test_coordinate_descent_inside_the_step_settles_its_iteration_count(dt):
    """Inside the dictionary step a coordinate-descent solve no longer reads its stop flag between the check sweep and
    the nine sweeps behind it: the device skips them when the test was met, and *lasso_it is settled at the end of the
    step (lasso_settle_deferred).  Both outcomes -- met at sweep 0 (warm start on converged codes: it = 0, codes those
    after ONE sweep) and not met (it = lasso_iter - 1) -- must give the codes and the count of the standalone solve
    (decomp_amd.lasso.solve, itself pinned against the oracle and the reference fixtures), through the blocking and
    the asynchronous entry."""
    import ctypes
    import torch
    import decomp_amd as decomp
    from decomp_amd import _arrays, _hip
    tdt = getattr(torch, dt)
    cplx = dt.startswith('complex')
    sfx = {'float32': 'f32', 'float64': 'f64', 'complex64': 'c64'}[dt]
    rdt = torch.float64 if dt == 'float64' else torch.float32
    g = torch.Generator(device='cuda')
    g.manual_seed(11)
    MB, F, K = 192, 96, 40

    def randn(*sh):
        r = torch.randn(sh, generator=g, device='cuda', dtype=rdt)
        return torch.complex(r, torch.randn(sh, generator=g, device='cuda', dtype=rdt)) if cplx else r
    D = randn(K, F)
    _arrays.l2_normalize_(D, strict=True)
    xt = randn(MB, K) * (torch.rand((MB, K), generator=g, device='cuda') < 0.15).to(rdt)
    Y = (xt @ D + 0.05 * randn(MB, F)).contiguous()
    lib, h = _arrays.lib_handle(D)
    tol = 1e-3 if dt != 'float64' else 1e-7
    alpha = 0.002           # (the solver scales it by n_channels: lasso.py:135-138)
    # converged codes to warm-start from
    it_c, x_conv = decomp.lasso.solve(Y, D, alpha, tol=tol, method='cd', maxiter=500)
    assert 0 < it_c < 499 and int((x_conv != 0).sum()) > 0
    for x0, want_met in ((x_conv, True), (torch.ones((MB, K), device='cuda', dtype=tdt), False)):
        it_ref, x_ref = decomp.lasso.solve(Y, D, alpha, x=x0.clone(), tol=tol, method='cd', maxiter=10)
        assert (it_ref == 0) == want_met
        for entry in ('dcp_dict_step_', 'dcp_dict_step_async_'):
            x = x0.clone()
            A = torch.zeros((K, K), device='cuda', dtype=tdt)
            B = torch.zeros((K, F), device='cuda', dtype=tdt)
            Dn = torch.empty_like(D)
            lit = ctypes.c_int(-7)
            if entry.endswith('async_'):
                md = torch.zeros((1,), device='cuda', dtype=rdt)
                out = _arrays.ptr(md)
            else:
                mdc = ctypes.c_double(0)
                out = ctypes.byref(mdc)
            _hip.check(h, getattr(lib, entry + sfx)(
                h, _arrays.ptr(Y), _arrays.ptr(x), _arrays.ptr(D), _arrays.ptr(Dn), _arrays.ptr(A), _arrays.ptr(B),
                MB, F, K, (1.0 - MB) / 1.0, alpha, _hip.LASSO_CD, 10, tol, out, ctypes.byref(lit)), entry)
            torch.cuda.synchronize()
            assert lit.value == it_ref, (dt, entry, want_met, lit.value, it_ref)
            assert torch.equal(x, x_ref), (dt, entry, want_met)
            assert bool(torch.isfinite(torch.view_as_real(Dn) if cplx else Dn).all())
