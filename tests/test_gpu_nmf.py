"""GPU parity: decomp_amd.nmf (HIP, through the C ABI) against the CPU oracle and the
golden vectors of the real reference."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _golden():
    return np.load(os.path.join(GOLDEN, 'nmf_golden.npz'), allow_pickle=False)


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))) /
                 max(1e-300, float(np.max(np.abs(b)))))


def _trace_hip(y, D0, mask, lik, n):
    """n iterations on the GPU with the per-iteration residual, via the product's own
    host layer (tol = 0 never stops early)."""
    import torch
    from decomp_amd import _arrays, nmf as hnmf
    yd = _arrays.to_device(y)
    md = _arrays.to_device(mask)
    Dd = _arrays.to_device(D0, copy=True)
    _arrays.l2_normalize_(Dd, strict=True)
    xd = torch.ones((y.shape[0], D0.shape[0]), dtype=yd.dtype, device='cuda')
    trace = []
    it = hnmf._run_mu(yd, md, xd, Dd, hnmf._likelihood_code(lik), 0.0, n + 1, resid_trace=trace)
    assert it == n + 1
    return Dd.cpu().numpy(), xd.cpu().numpy(), np.array(trace)


@pytest.mark.parametrize('case', [
    'nmf_64x48k4', 'nmf_101x20k3', 'nmf_256x128k8'])
@pytest.mark.parametrize('dt', ['float64', 'float32'])
@pytest.mark.parametrize('lik', ['l2', 'kl'])
@pytest.mark.parametrize('mtag', ['nomask', 'mask'])
def test_trace_matches_reference(case, dt, lik, mtag):
    """25 iterations: per-iteration residual within 1e-5 rel (fp32) / 1e-10 (fp64) of the
    reference's own trace; final x, D likewise."""
    g = _golden()
    base = '%s_%s_%s' % (case, dt, lik)
    name = base + '_' + mtag
    y, D0 = g[base + '/y'], g[base + '/D0']
    mask = g[base + '/mask'] if mtag == 'mask' else None
    ref_res = g[name + '/trace_resid']
    D, x, res = _trace_hip(y, D0, mask, lik, len(ref_res))
    tol = 1e-5 if dt == 'float32' else 1e-10
    assert len(res) == len(ref_res)
    assert np.max(np.abs(res - ref_res) / ref_res) < tol, np.max(np.abs(res - ref_res) / ref_res)
    xt = 2e-4 if dt == 'float32' else 1e-9
    assert _rel(D, g[name + '/trace_D']) < xt
    assert _rel(x, g[name + '/trace_x']) < xt
    assert D.dtype == y.dtype and x.dtype == y.dtype


@pytest.mark.parametrize('dt', ['float64', 'float32'])
@pytest.mark.parametrize('lik', ['l2', 'kl'])
@pytest.mark.parametrize('mtag', ['nomask', 'mask'])
def test_public_solve_matches_reference(dt, lik, mtag):
    """decomp_amd.nmf.solve end to end on the reference test's own shape
    (tests/test_nmf.py:59-102): same iteration count (fp64) and same solution."""
    import decomp_amd
    g = _golden()
    base = 'nmf_101x20k3_%s_%s' % (dt, lik)
    name = base + '_' + mtag
    y, D0 = g[base + '/y'], g[base + '/D0']
    mask = g[base + '/mask'] if mtag == 'mask' else None
    it, D, x = decomp_amd.nmf.solve(y, D0.copy(), x=None, tol=float(g[name + '/tol']),
                                    maxiter=400, method='mu', likelihood=lik, mask=mask,
                                    random_seed=0)
    assert isinstance(D, np.ndarray) and D.dtype == y.dtype
    if dt == 'float64':
        assert it == int(g[name + '/it'])
        assert _rel(D, g[name + '/D']) < 1e-9 and _rel(x, g[name + '/x']) < 1e-9
    else:
        assert abs(it - int(g[name + '/it'])) <= 3
        assert _rel(D, g[name + '/D']) < 5e-4 and _rel(x, g[name + '/x']) < 5e-4


def test_against_oracle_medium_fp32():
    """8192 x 1024, k = 64 (SURVEY 8d parity gate): 20 iterations, residual per
    iteration within 1e-5 rel of the CPU oracle."""
    from oracle import nmf as onmf, common
    rng = np.random.RandomState(0)
    N, F, K = 8192, 1024, 64
    xt = np.maximum(rng.randn(N, K), 0).astype(np.float32)
    Dt = np.maximum(rng.randn(K, F), 0).astype(np.float32)
    y = (xt @ Dt + 0.1 * np.abs(rng.randn(N, F))).astype(np.float32)
    D0 = np.maximum(Dt + 0.3 * rng.randn(K, F), 0.1).astype(np.float32)
    n = 20
    D, x, res = _trace_hip(y, D0, None, 'l2', n)
    Do = common.l2_strict(D0)
    xo = np.ones((N, K), np.float32)
    for i in range(n):
        xo, Do, _ = onmf.mu_step(y, xo, Do)
        ro = onmf.residual(y, xo, Do)
        assert abs(res[i] - ro) / ro < 1e-5, (i, res[i], ro)
    assert _rel(D, Do) < 1e-3 and _rel(x, xo) < 1e-3


def test_masked_entries_contribute_exactly_zero():
    """Values under the mask must not influence anything: flip them to garbage and
    require bit-identical outputs."""
    import decomp_amd
    rng = np.random.RandomState(1)
    N, F, K = 300, 96, 5
    y = np.abs(rng.randn(N, F)).astype(np.float32)
    D0 = np.abs(rng.randn(K, F)).astype(np.float32) + 0.1
    mask = (rng.uniform(size=(N, F)) >= 0.2).astype(np.float32)
    y2 = y.copy()
    y2[mask == 0] = 1.0e3 * rng.rand(int((mask == 0).sum())).astype(np.float32)
    for lik in ('l2', 'kl'):
        a = decomp_amd.nmf.solve(y, D0.copy(), tol=0.0, maxiter=12, likelihood=lik, mask=mask)
        b = decomp_amd.nmf.solve(y2, D0.copy(), tol=0.0, maxiter=12, likelihood=lik, mask=mask)
        assert a[0] == b[0] == 12
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_torch_tensors_in_place_semantics_and_errors():
    import torch
    import decomp_amd
    rng = np.random.RandomState(2)
    y = torch.from_numpy(np.abs(rng.randn(64, 32)).astype(np.float32)).cuda()
    D0 = torch.from_numpy(np.abs(rng.randn(4, 32)).astype(np.float32) + 0.1).cuda()
    D0_copy = D0.clone()
    it, D, x = decomp_amd.nmf.solve(y, D0, tol=1e-3, maxiter=50)
    assert isinstance(D, torch.Tensor) and D.is_cuda and x.shape == (64, 4)
    assert torch.equal(D0, D0_copy)          # inputs are not mutated
    # negative D -> AssertionError (assertion.py:99-100)
    Dneg = D0.clone(); Dneg[0, 0] = -1.0
    with pytest.raises(AssertionError):
        decomp_amd.nmf.solve(y, Dneg)
    # mixing array kinds -> TypeError (cp_compat.py:13)
    with pytest.raises(TypeError):
        decomp_amd.nmf.solve(y.cpu().numpy(), D0)
    with pytest.raises(NotImplementedError):
        decomp_amd.nmf.solve(y, D0, method='nope')


def _sharded_problem():
    rng = np.random.RandomState(7)
    N, F, K = 768, 1536, 24
    xt = np.maximum(rng.randn(N, K), 0).astype(np.float32)
    Dt = np.maximum(rng.randn(K, F), 0).astype(np.float32)
    y = (xt @ Dt + 0.1 * np.abs(rng.randn(N, F))).astype(np.float32)
    D0 = np.maximum(Dt + 0.3 * rng.randn(K, F), 0.1).astype(np.float32)
    mask = (rng.uniform(size=(N, F)) >= 0.2).astype(np.float32)
    return y, D0, mask


@pytest.mark.parametrize('masked', [False, True])
def test_sharded_driver_world1_equals_solve(masked):
    """decomp_amd.sharded (stats -> [all-reduce] -> update, speculative next iteration with
    rollback) on one rank must reproduce nmf.solve exactly, including the stop iteration."""
    import torch
    import decomp_amd
    from decomp_amd import sharded
    y, D0, mask = _sharded_problem()
    m = mask if masked else None
    it, D, x = decomp_amd.nmf.solve(y, D0.copy(), tol=2e-3, maxiter=200, mask=m)
    assert 2 < it < 199
    its, Ds, xs = sharded.nmf_solve_sharded(torch.from_numpy(y).cuda(), torch.from_numpy(D0).cuda(),
                                            tol=2e-3, maxiter=200,
                                            mask_local=None if m is None else torch.from_numpy(m).cuda())
    assert its == it
    assert np.array_equal(Ds.cpu().numpy(), D) and np.array_equal(xs.cpu().numpy(), x)


def _gloo_gpu_worker(rank, world, port, q, masked=False):
    import os
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from decomp_amd import sharded
        y, D0, mask = _sharded_problem()
        rows = slice(rank * len(y) // world, (rank + 1) * len(y) // world)
        it, D, x = sharded.nmf_solve_sharded(torch.from_numpy(y[rows]).cuda(),
                                             torch.from_numpy(D0).cuda(), tol=2e-3, maxiter=200,
                                             mask_local=torch.from_numpy(mask[rows]).cuda() if masked else None)
        # round 4: the loop that ran is dcp_nmf_mu_sharded_* -- the in-library loop a multi-GPU rank executes -- with
        # the exchange handed in as a callback over gloo (dcp_comm_set_external), not the Python loop
        assert sharded.communicator_kind(D) == 'external'
        q.put((rank, it, D.cpu().numpy(), x.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('masked', [False, True])
def test_sharded_two_ranks_on_one_gpu_gloo(masked):
    """Two processes sharing the one GPU of the test box, statistics all-reduced over gloo: the IN-LIBRARY sharded
    loop (dcp_nmf_mu_sharded_f32: statistics -> exchange -> replicated update, lagged stop test, discarded speculative
    iteration) with world_size = 2 -- RCCL refuses two ranks on one GPU, so the exchange is the library's external
    callback (masked: the [K, 2F] numerator | denominator statistics of configs[3]'s form cross the all-reduce)."""
    import os
    import torch.multiprocessing as mp
    import decomp_amd
    y, D0, mask = _sharded_problem()
    it1, D1, x1 = decomp_amd.nmf.solve(y, D0.copy(), tol=2e-3, maxiter=200, mask=mask if masked else None)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000) + (1000 if masked else 0)
    procs = [ctx.Process(target=_gloo_gpu_worker, args=(r, 2, port, q, masked)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]
    assert abs(res[0][1] - it1) <= 1
    assert np.array_equal(res[0][2], res[1][2])            # replicated D identical on both ranks
    x_all = np.concatenate([res[0][3], res[1][3]], axis=0)
    if res[0][1] == it1:
        assert _rel(res[0][2], D1) < 1e-4 and _rel(x_all, x1) < 1e-3


def _rccl_world1_worker(port, q):
    import os
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        from decomp_amd import _hip, sharded
        y, D0, _ = _sharded_problem()
        Y = torch.from_numpy(y).cuda()
        D = torch.from_numpy(D0).cuda()
        from decomp_amd import _arrays
        _arrays.l2_normalize_(D, strict=True)
        x = torch.ones((Y.shape[0], D.shape[0]), dtype=torch.float32, device='cuda')
        backend = sharded.HipStepBackend(Y, None, x, D, _hip.LIK_L2)
        # world_size=2 forces the collective call; on a 1-rank RCCL group it is the identity
        it, Dout = sharded.mu_loop(backend, D, 2e-3, 200, world_size=2, new_like=torch.empty_like)
        torch.cuda.synchronize()
        q.put((it, Dout.cpu().numpy(), backend.x.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_sharded_loop_over_rccl_world1():
    """The step loop with the REAL RCCL backend ("nccl") on the one GPU of the test box: the
    all-reduce is issued every iteration on a 1-rank communicator (identity), which exercises
    the stream ordering between the library's kernels and the collective; result must equal
    nmf.solve bit for bit."""
    import os
    import torch.multiprocessing as mp
    import decomp_amd
    y, D0, _ = _sharded_problem()
    it1, D1, x1 = decomp_amd.nmf.solve(y, D0.copy(), tol=2e-3, maxiter=200)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000)
    p = ctx.Process(target=_rccl_world1_worker, args=(port, q))
    p.start()
    it, D, x = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert it == it1
    assert np.array_equal(D, D1) and np.array_equal(x, x1)


def _in_library_world1_worker(q, dt, masked, lik):
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    torch.cuda.set_device(0)
    from decomp_amd import _arrays, _hip, sharded
    y, D0, mask = _sharded_problem()
    tdt = torch.float32 if dt == 'f32' else torch.float64
    Y = torch.from_numpy(y).cuda().to(tdt)
    M = torch.from_numpy(mask).cuda().to(tdt) if masked else None
    D = torch.from_numpy(D0).cuda().to(tdt)
    _arrays.l2_normalize_(D, strict=True)
    x = torch.ones((Y.shape[0], D.shape[0]), dtype=tdt, device='cuda')
    assert sharded.attach_communicator(D), 'RCCL communicator could not be created on the GPU box'
    lib, h = _arrays.lib_handle(D)
    import ctypes
    r, w = ctypes.c_int(-1), ctypes.c_int(-1)
    lib.dcp_comm_info(h, ctypes.byref(r), ctypes.byref(w))
    assert (r.value, w.value) == (0, 1)
    it = sharded.mu_solve_in_library(Y, M, x, D, _hip.LIK_KL if lik == 'kl' else _hip.LIK_L2, 2e-3, 200)
    # the collective entry point on complex data (configs[4]'s statistics dtype): identity on one rank
    c = torch.complex(torch.randn(37, 11, device='cuda'), torch.randn(37, 11, device='cuda'))
    c0 = c.clone()
    sharded.comm_allreduce_(c)
    torch.cuda.synchronize()
    assert torch.equal(c, c0)
    sharded.detach_communicator(D)
    lib.dcp_comm_info(h, ctypes.byref(r), ctypes.byref(w))
    assert w.value == 0
    q.put((it, D.cpu().numpy(), x.cpu().numpy()))


@pytest.mark.parametrize('dt,masked,lik', [('f32', False, 'l2'), ('f32', True, 'l2'), ('f64', False, 'l2'),
                                           ('f32', False, 'kl')])
def test_in_library_sharded_loop_over_rccl_world1(dt, masked, lik):
    """dcp_nmf_mu_sharded_*: the loop a rank of a multi-GPU run executes -- statistics, ncclAllReduce on the
    handle's own stream (a 1-rank RCCL communicator created through dcp_comm_unique_id / dcp_comm_init: the
    identity), replicated update, lagged stop test -- must reproduce nmf.solve bit for bit, stop iteration
    included.  batch_mu.py:16-26."""
    import torch.multiprocessing as mp
    import decomp_amd
    y, D0, mask = _sharded_problem()
    npdt = np.float32 if dt == 'f32' else np.float64
    it1, D1, x1 = decomp_amd.nmf.solve(y.astype(npdt), D0.astype(npdt), tol=2e-3, maxiter=200,
                                       mask=mask.astype(npdt) if masked else None, likelihood=lik)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_in_library_world1_worker, args=(q, dt, masked, lik))
    p.start()
    it, D, x = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert it == it1 and 2 < it < 199
    assert np.array_equal(D, D1) and np.array_equal(x, x1)


def test_sharded_entry_without_communicator_is_an_error():
    import ctypes
    import torch
    from decomp_amd import _arrays, _hip
    D = torch.rand(4, 32, device='cuda')
    Y = torch.rand(16, 32, device='cuda')
    x = torch.ones(16, 4, device='cuda')
    lib, h = _arrays.lib_handle(D)
    it = ctypes.c_int(0)
    rc = lib.dcp_nmf_mu_sharded_f32(h, _arrays.ptr(Y), None, _arrays.ptr(x), _arrays.ptr(D), 16, 32, 4, 0,
                                    ctypes.c_float(0.0), 3, ctypes.byref(it), None)
    assert rc == _hip.ERR_COMM
    assert lib.dcp_comm_allreduce_sum_f32(h, _arrays.ptr(D), 8) == _hip.ERR_COMM


@pytest.mark.parametrize('shape', [(1000, 1200, 24), (4096, 1024, 256), (333, 77, 5)])
def test_mask_row_bits_identical_to_float_mask_and_fallback(shape):
    """The masked forward product multiplies by row bits fetched ahead of the GEMM main loop when the
    float32 mask is 0/1 (dcp_nmf_mask_prepare_*): results must be BIT-identical to the float-mask
    epilogue; a fractional mask must report binary = 0 and match the oracle through the float path."""
    import ctypes
    import torch
    from decomp_amd import _arrays, _hip
    from oracle import nmf as onmf, common
    N, F, K = shape
    rng = np.random.RandomState(N)
    xt = np.maximum(rng.randn(N, K), 0)
    Dt = np.maximum(rng.randn(K, F), 0)
    y = (xt @ Dt + 0.1 * np.abs(rng.randn(N, F))).astype(np.float32)
    D0 = common.l2_strict(np.maximum(Dt + 0.3 * rng.randn(K, F), 0.1).astype(np.float32))
    mask = (rng.uniform(size=(N, F)) >= 0.2).astype(np.float32)
    Y, M, D = torch.from_numpy(y).cuda(), torch.from_numpy(mask).cuda(), torch.from_numpy(D0).cuda()
    lib, h = _arrays.lib_handle(Y)
    W = lib.dcp_nmf_mu_stats_width(F, K, 0, 1)

    def prepared(Mt, want_bits):
        Ym = torch.empty_like(Y)
        bits = torch.empty((lib.dcp_nmf_mask_bits_words(N, F),), dtype=torch.int32, device='cuda')
        binary = ctypes.c_int(-1)
        _hip.check(h, lib.dcp_nmf_mask_prepare_f32(h, _arrays.ptr(Y), _arrays.ptr(Mt), N, F, _arrays.ptr(Ym),
                                                   _arrays.ptr(bits), ctypes.byref(binary)), 'prepare')
        x = torch.ones((N, K), device='cuda')
        xo, stats = torch.empty_like(x), torch.empty((K, W), device='cuda')
        _hip.check(h, lib.dcp_nmf_mu_stats_prepared_f32(
            h, _arrays.ptr(Ym), _arrays.ptr(Mt), _arrays.ptr(bits) if want_bits else None, _arrays.ptr(x),
            _arrays.ptr(xo), _arrays.ptr(D), N, F, K, 0, _arrays.ptr(stats)), 'stats_prepared')
        torch.cuda.synchronize()
        return binary.value, xo, stats
    b1, x1, s1 = prepared(M, True)
    b0, x0, s0 = prepared(M, False)
    assert b1 == 1 and b0 == 1
    assert torch.equal(x1, x0) and torch.equal(s1, s0)
    # the un-prepared entry point gives the same numbers
    x = torch.ones((N, K), device='cuda')
    xo, stats = torch.empty_like(x), torch.empty((K, W), device='cuda')
    _hip.check(h, lib.dcp_nmf_mu_stats_f32(h, _arrays.ptr(Y), _arrays.ptr(M), _arrays.ptr(x), _arrays.ptr(xo),
                                           _arrays.ptr(D), N, F, K, 0, _arrays.ptr(stats)), 'stats')
    assert torch.equal(xo, x1) and torch.equal(stats, s1)
    # fractional mask: not binary -> float path, against the oracle
    frac = (mask * rng.uniform(0.25, 1.0, size=mask.shape)).astype(np.float32)
    bf, xf, sf = prepared(torch.from_numpy(frac).cuda(), True)
    assert bf == 0
    import decomp_amd
    it, Dg, xg = decomp_amd.nmf.solve(y, D0.copy(), tol=0.0, maxiter=4, mask=frac)
    ito, Do, xo_ = onmf.solve(y, D0.copy(), tol=0.0, maxiter=4, mask=frac)
    assert np.max(np.abs(Dg - Do)) < 2e-5 and np.max(np.abs(xg - xo_)) < 2e-4 * np.max(np.abs(xo_))
