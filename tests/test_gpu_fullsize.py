"""GPU: BASELINE.json's full sizes.  Primary evidence: the NumPy oracle itself iterated at the
full configs[1] shape (Y 65536 x 4096, k = 256, float32) and at one 16384-row masked shard of
configs[3] for three MU iterations, per-iteration residual within 1e-5 relative (north_star).
Then size-independent properties and fp32-vs-fp64 traces.  Data are synthesised on the GPU with
torch (test plumbing only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, F, K = 65536, 4096, 256


def _data(rows, seed=0):
    import torch
    g = torch.Generator(device='cuda')
    g.manual_seed(seed)
    Dt = torch.randn((K, F), generator=g, device='cuda').clamp_(min=0)
    xt = torch.randn((rows, K), generator=g, device='cuda').clamp_(min=0)
    Y = xt @ Dt + 0.1 * torch.randn((rows, F), generator=g, device='cuda').abs_()
    D0 = (Dt + 0.3 * torch.randn((K, F), generator=g, device='cuda')).clamp_(min=0.1)
    return Y, D0


def _resid(Y, x, D, mask=None):
    from decomp_amd import _arrays, _hip
    import ctypes
    lib, h = _arrays.lib_handle(Y)
    out = ctypes.c_double(0)
    _hip.check(h, lib.dcp_nmf_residual_f32(h, _arrays.ptr(Y), _arrays.ptr(mask), _arrays.ptr(x),
                                           _arrays.ptr(D), Y.shape[0], F, K, ctypes.byref(out)), 'resid')
    return out.value


def _host_residual(y, x, d, mask=None, block=8192):
    """||(y - x d) o mask||_F, float64 accumulation, on the host in row blocks: the SAME function
    scores the HIP path's and the oracle's iterates."""
    acc = 0.0
    for r0 in range(0, y.shape[0], block):
        r = y[r0:r0 + block] - x[r0:r0 + block].dot(d)
        if mask is not None:
            r = r * mask[r0:r0 + block]
        acc += float(np.sum(np.square(r, dtype=np.float64)))
    return acc ** 0.5


def _oracle_vs_hip(rows, seed, masked, n_it=3):
    """n_it MU iterations of the NumPy oracle (oracle.nmf.mu_step = the reference's formulation,
    grads.py:108-125 + batch_mu.py:16-24) and of the HIP path from the same (Y, D0, x = ones);
    returns the per-iteration relative residual differences and max|D - D_oracle|."""
    import ctypes
    import torch
    from decomp_amd import _arrays, _hip
    from oracle import nmf as onmf, common
    Y, D0 = _data(rows, seed=seed)
    mask = None
    if masked:
        g = torch.Generator(device='cuda')
        g.manual_seed(seed + 100)
        mask = (torch.rand((rows, F), generator=g, device='cuda') >= 0.2).float()
    y, d0 = Y.cpu().numpy(), D0.cpu().numpy()
    m = None if mask is None else mask.cpu().numpy()
    xg = torch.ones((rows, K), device='cuda')
    Dg = D0.clone()
    _arrays.l2_normalize_(Dg, strict=True)
    lib, h = _arrays.lib_handle(Y)
    it = ctypes.c_int(0)
    x, d = np.ones((rows, K), np.float32), common.l2_strict(d0)
    rel, ddiff = [], []
    for _ in range(n_it):
        _hip.check(h, lib.dcp_nmf_mu_f32(h, _arrays.ptr(Y), _arrays.ptr(mask), _arrays.ptr(xg), _arrays.ptr(Dg),
                                         rows, F, K, _hip.LIK_L2, ctypes.c_float(0.0), 2, ctypes.byref(it),
                                         None, None), 'dcp_nmf_mu_f32')
        x, d, _ = onmf.mu_step(y, x, d, m)
        r_cpu = _host_residual(y, x, d, m)
        r_hip = _host_residual(y, xg.cpu().numpy(), Dg.cpu().numpy(), m)
        rel.append(abs(r_hip - r_cpu) / r_cpu)
        ddiff.append(float(np.max(np.abs(Dg.cpu().numpy() - d))))
    return rel, ddiff


def test_c2_oracle_parity_full_shape():
    """north_star: "per-iteration residual matching NumPy to 1e-5 rel" at the FULL configs[1] shape,
    against the NumPy oracle itself (SURVEY 8d "parity gates ... at C2 for 3 iterations")."""
    rel, ddiff = _oracle_vs_hip(N, seed=21, masked=False)
    assert max(rel) <= 1e-5, rel
    assert max(ddiff) <= 1e-5, ddiff          # unit-norm rows: absolute = relative scale


def test_c4_shard_oracle_parity_masked():
    """The same gate for one 16384-row shard of the masked configs[3] (20 % missing)."""
    rel, ddiff = _oracle_vs_hip(16384, seed=23, masked=True)
    assert max(rel) <= 1e-5, rel
    assert max(ddiff) <= 1e-5, ddiff


def test_c2_shard_oracle_parity():
    """One 8192-row shard of configs[1]: the split Y.D^T product with the quotient in the x.G epilogue."""
    rel, ddiff = _oracle_vs_hip(8192, seed=29, masked=False)
    assert max(rel) <= 1e-5, rel
    assert max(ddiff) <= 1e-5, ddiff


def test_c2_properties():
    import torch
    import decomp_amd
    Y, D0 = _data(N)
    res = []
    Dk, xk = D0, None
    for n_it in (2, 4, 7):          # warm restarts: (it, D, x) of n_it - 1 more iterations each
        it, Dk, xk = decomp_amd.nmf.solve(Y, Dk, x=xk, tol=0.0, maxiter=n_it)
        assert it == n_it
        res.append(_resid(Y, xk, Dk))
    # MU never increases the l2 loss
    assert res[0] >= res[1] >= res[2] > 0
    # rows of D are unit norm, everything non-negative and finite
    nr = torch.linalg.vector_norm(Dk, dim=1)
    assert float((nr - 1).abs().max()) < 1e-5
    assert bool((Dk >= 0).all()) and bool((xk >= 0).all()) and bool(torch.isfinite(xk).all())

    # homogeneity: scaling Y by c scales x by c and leaves the normalised D unchanged
    it1, D1, x1 = decomp_amd.nmf.solve(Y, D0, tol=0.0, maxiter=4)
    it2, D2, x2 = decomp_amd.nmf.solve(Y * 4.0, D0, tol=0.0, maxiter=4)
    assert float((D1 - D2).abs().max()) < 1e-5
    assert float((x2 - 4.0 * x1).abs().max()) <= 1e-4 * float(x1.abs().max()) * 4.0

    # row-permutation equivariance: shuffling the samples permutes x and leaves D unchanged
    # (up to the summation order of the split-K statistics)
    perm = torch.randperm(N, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
    it3, D3, x3 = decomp_amd.nmf.solve(Y[perm].contiguous(), D0, tol=0.0, maxiter=4)
    assert float((D3 - D1).abs().max()) < 1e-5
    assert float((x3 - x1[perm]).abs().max()) <= 1e-4 * float(x1.abs().max())

    # run-to-run bitwise reproducibility (ordered slab sums, no float atomics)
    it4, D4, x4 = decomp_amd.nmf.solve(Y, D0, tol=0.0, maxiter=4)
    assert torch.equal(D4, D1) and torch.equal(x4, x1)


def test_c2_sharded_statistics_equal_unsharded():
    """Sum of the per-shard statistics == statistics of the whole (what the all-reduce relies
    on), at the full C2 shape with 8 shards of 8192 rows."""
    import ctypes
    import torch
    from decomp_amd import _arrays, _hip
    Y, D0 = _data(N, seed=3)
    D = D0.clone()
    _arrays.l2_normalize_(D, strict=True)
    lib, h = _arrays.lib_handle(Y)
    W = F + K
    x = torch.ones((N, K), device='cuda')
    xo = torch.empty_like(x)
    whole = torch.empty((K, W), device='cuda')
    _hip.check(h, lib.dcp_nmf_mu_stats_f32(h, _arrays.ptr(Y), None, _arrays.ptr(x), _arrays.ptr(xo),
                                           _arrays.ptr(D), N, F, K, 0, _arrays.ptr(whole)), 'stats')
    acc = torch.zeros((K, W), device='cuda', dtype=torch.float64)
    part = torch.empty((K, W), device='cuda')
    rows = N // 8
    for s in range(8):
        ys, xs = Y[s * rows:(s + 1) * rows], x[s * rows:(s + 1) * rows]
        xs_o = torch.empty_like(xs)
        _hip.check(h, lib.dcp_nmf_mu_stats_f32(h, _arrays.ptr(ys), None, _arrays.ptr(xs), _arrays.ptr(xs_o),
                                               _arrays.ptr(D), rows, F, K, 0, _arrays.ptr(part)), 'stats')
        acc += part.double()
        assert float((xs_o - xo[s * rows:(s + 1) * rows]).abs().max()) <= 2e-5 * float(xo.abs().max())
    rel = float((acc - whole.double()).abs().max() / whole.double().abs().max())
    assert rel < 1e-5, rel


def test_c4_shard_masked_properties():
    """One 16384-row shard of configs[3] (20 % missing): masked entries contribute exactly
    zero at full width, the masked residual decreases."""
    import torch
    import decomp_amd
    rows = 16384
    Y, D0 = _data(rows, seed=5)
    g = torch.Generator(device='cuda')
    g.manual_seed(9)
    mask = (torch.rand((rows, F), generator=g, device='cuda') >= 0.2).float()
    garbage = Y.clone()
    garbage[mask == 0] = 1.0e4
    a = decomp_amd.nmf.solve(Y, D0, tol=0.0, maxiter=4, mask=mask)
    b = decomp_amd.nmf.solve(garbage, D0, tol=0.0, maxiter=4, mask=mask)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    r1 = _resid(Y, a[2], a[1], mask)
    c = decomp_amd.nmf.solve(Y, a[1], x=a[2], tol=0.0, maxiter=4, mask=mask)
    assert _resid(Y, c[2], c[1], mask) <= r1


def _dl_data(rows, F_, K_, cplx, seed=2):
    """SURVEY 8d C3 / C5 recipe (amplitude 30 at 5 % density so that alpha * F leaves a sparse,
    non-empty code)."""
    import torch
    g = torch.Generator(device='cuda')
    g.manual_seed(seed)

    def randn(*s):
        r = torch.randn(s, generator=g, device='cuda')
        return torch.complex(r, torch.randn(s, generator=g, device='cuda')) if cplx else r
    Dt = randn(K_, F_)
    xt = 30.0 * randn(rows, K_) * (torch.rand((rows, K_), generator=g, device='cuda') < 0.05)
    Y = xt @ Dt + 0.1 * randn(rows, F_)
    D0 = Dt + 0.2 * randn(K_, F_)
    return Y, D0


def _dl_objective(Y, x, D, alpha):
    """Mean over rows of 1/(2 n) |y - x D|^2 + alpha |x|_1 with row-normalised D (the quantity
    dictionary learning decreases, tests/test_dictionary.py:10-16)."""
    import torch
    Dn = D / torch.sqrt(torch.clamp(torch.sum(torch.abs(D) ** 2, dim=1, keepdim=True), min=1.0))
    r = Y - x @ Dn
    return float((0.5 / Y.shape[1] * torch.sum(torch.abs(r) ** 2) + alpha * torch.sum(torch.abs(x))) / Y.shape[0])


def _dl_host_data(rows, F_, K_, cplx, seed):
    """SURVEY 8(d) C3 / C5 recipe on the HOST with np.random.RandomState (the data the oracle and the
    HIP path both start from): Dt = randn(K, F), xt = 30 randn . Bernoulli(0.05), Y = xt Dt + 0.1 randn,
    D0 = Dt + 0.2 randn; complex: randn + 1j randn, complex64."""
    rng = np.random.RandomState(seed)
    dt = np.complex64 if cplx else np.float32

    def randn(*s):
        r = rng.randn(*s).astype(np.float32)
        return (r + 1j * rng.randn(*s).astype(np.float32)).astype(dt) if cplx else r
    Dt = randn(K_, F_)
    xt = (30.0 * randn(rows, K_) * (rng.uniform(size=(rows, K_)) < 0.05)).astype(dt)
    Y = (xt @ Dt + np.float32(0.1) * randn(rows, F_)).astype(dt)
    D0 = (Dt + np.float32(0.2) * randn(K_, F_)).astype(dt)
    return Y, D0


def _hip_dict_steps(Y, D0, x0, alpha, method_code, lasso_iter, lasso_tol, n_steps):
    """n_steps calls of dcp_dict_step_* on the same minibatch (as the reference's epochs revisit it):
    returns per step (lasso_it, x, A, B, D_new, maxdiff) as host arrays."""
    import ctypes
    import torch
    from decomp_amd import _arrays, _hip
    Yd = torch.from_numpy(Y).cuda()
    D = torch.from_numpy(D0).cuda()
    _arrays.l2_normalize_(D, strict=True)                    # dictionary_learning.py:126
    x = torch.from_numpy(x0).cuda()
    Nb, F_ = Y.shape
    K_ = D0.shape[0]
    A = torch.zeros((K_, K_), dtype=D.dtype, device='cuda')
    B = torch.zeros((K_, F_), dtype=D.dtype, device='cuda')
    Dn = torch.empty_like(D)
    lib, h = _arrays.lib_handle(D)
    step = getattr(lib, 'dcp_dict_step_' + _arrays.suffix(D))
    out = []
    md, lit = ctypes.c_double(0), ctypes.c_int(0)
    for count in range(n_steps):
        theta = count * Nb + 1.0
        _hip.check(h, step(h, _arrays.ptr(Yd), _arrays.ptr(x), _arrays.ptr(D), _arrays.ptr(Dn), _arrays.ptr(A),
                           _arrays.ptr(B), Nb, F_, K_, (theta - Nb) / theta, alpha, method_code, lasso_iter,
                           lasso_tol, ctypes.byref(md), ctypes.byref(lit)), 'dcp_dict_step')
        out.append((lit.value, x.cpu().numpy(), A.cpu().numpy(), B.cpu().numpy(), Dn.cpu().numpy(), md.value))
        D, Dn = Dn, D
    return out


def _rel(a, b):
    return float(np.max(np.abs(a - b))) / max(float(np.max(np.abs(b))), 1e-30)


@pytest.mark.parametrize('cplx,F_', [(False, 4096), (True, 8192)])
def test_dictionary_step_oracle_parity_ista_at_timed_shape(cplx, F_):
    """The dictionary step exactly as bench.py times it -- one minibatch of BASELINE configs[2]
    (8192 x 4096, k = 512, float32) / configs[4] (8192 x 8192, k = 512, complex64), alpha = 0.1, ista x 10
    -- against the NumPy oracle on the same host data, two consecutive steps (the second with beta > 0,
    non-zero A, B and warm codes): oracle.dictionary_learning.minibatch_step = lasso.solve_fastpath
    (lasso.py:97-189, 274-297) + A, B accumulation (dictionary_learning.py:143-152) + the sequential atom
    sweep (:154-159).  Identical LASSO iteration count; x, A, B, D_new within 2e-4 of the largest entry
    (single precision, K = 512 dependent atom updates)."""
    from decomp_amd import _hip
    from oracle import dictionary_learning as odl
    from oracle.common import l2_strict
    rows, K_ = 8192, 512
    Y, D0 = _dl_host_data(rows, F_, K_, cplx, seed=3 if cplx else 2)
    x0 = np.ones((rows, K_), dtype=Y.dtype)                  # dictionary_learning.py:58-59
    hip = _hip_dict_steps(Y, D0, x0, 0.1, _hip.LASSO_ISTA, 10, 1e-5, 2)
    D = l2_strict(D0)
    A = np.zeros((K_, K_), Y.dtype)
    B = np.zeros((K_, F_), Y.dtype)
    x = x0
    for count in range(2):
        it2, x, A, B, D_new, diff = odl.minibatch_step(Y, x, D, A, B, count, rows, 0.1, 'ista', 10, 1e-5)
        g_it, gx, gA, gB, gD, gdiff = hip[count]
        assert g_it == it2, (count, g_it, it2)
        dens = float((x != 0).mean())
        assert 0.005 < dens < 0.3, dens                      # a sparse, non-empty code
        assert _rel(gx, x) < 2e-4, (count, 'x', _rel(gx, x))
        assert _rel(gA, A) < 2e-4, (count, 'A', _rel(gA, A))
        assert _rel(gB, B) < 2e-4, (count, 'B', _rel(gB, B))
        assert _rel(gD, D_new) < 2e-4, (count, 'D', _rel(gD, D_new))
        assert abs(gdiff - diff) <= 2e-4 * max(1.0, diff), (count, gdiff, diff)
        D = D_new


def test_dictionary_step_oracle_parity_cd_at_timed_shape():
    """The same step with the reference's DEFAULT inner solver (lasso_method = 'cd',
    dictionary_learning.py:14).  The reference's sweep recomputes x.A for every coordinate
    (lasso.py:539-551: 2 N K F flops per coordinate, 1.8e14 for this minibatch), so the oracle runs
    stage-wise: (1) its as-written coordinate descent on 64 rows spread over the minibatch (rows are
    independent given D; from x = 1 the all-rows stop test at sweep 0 cannot fire, so both run the full 10
    sweeps) pins the codes; (2) A, B and the sequential atom sweep of the oracle, fed with the HIP path's
    codes, pin the D side at K = 512, F = 4096."""
    from decomp_amd import _hip
    from oracle import dictionary_learning as odl, lasso as olasso
    from oracle.common import l2_strict
    rows, F_, K_ = 8192, 4096, 512
    Y, D0 = _dl_host_data(rows, F_, K_, False, seed=2)
    x0 = np.ones((rows, K_), dtype=np.float32)
    (g_it, gx, gA, gB, gD, gdiff), = _hip_dict_steps(Y, D0, x0, 0.1, _hip.LASSO_CD, 10, 1e-5, 1)
    D = l2_strict(D0)
    sel = np.arange(0, rows, rows // 64)
    it2, xs = olasso.solve_fastpath(Y[sel], D, 0.1, x=x0[sel].copy(), tol=1e-5, maxiter=10, method='cd')
    assert g_it == it2 == 9
    assert 0.005 < float((xs != 0).mean()) < 0.3
    assert _rel(gx[sel], xs) < 2e-4, _rel(gx[sel], xs)
    beta = (1.0 - rows) / 1.0
    A = beta * np.zeros((K_, K_), np.float32) + gx.T @ gx
    B = beta * np.zeros((K_, F_), np.float32) + gx.T @ Y
    D_new = odl.atom_sweep(D, A, B)
    assert _rel(gA, A) < 2e-4 and _rel(gB, B) < 2e-4
    assert _rel(gD, D_new) < 2e-4, _rel(gD, D_new)
    assert abs(gdiff - float(np.max(np.abs(D - D_new)))) <= 2e-4


@pytest.mark.parametrize('lasso_method', ['ista', 'cd'])
def test_c3_dictionary_learning_properties(lasso_method):
    """BASELINE configs[2] at its full shape (Y 65536 x 4096, k = 512, alpha = 0.1, fp32, minibatch 8192),
    two epochs over 8 minibatches: finite unit-norm-bounded atoms, sparse non-empty codes, identical result
    on a re-run (deterministic split-K and shuffle), and a lower objective than the starting point."""
    import torch
    import decomp_amd
    rows, F_, K_ = 65536, 4096, 512
    Y, D0 = _dl_data(rows, F_, K_, cplx=False)
    kw = dict(tol=0.0, minibatch=8192, maxiter=3, lasso_method=lasso_method, lasso_iter=10,
              lasso_tol=1e-5, random_seed=0)
    it, D, x = decomp_amd.dictionary_learning.solve(Y, D0, 0.1, **kw)
    it2, D2, x2 = decomp_amd.dictionary_learning.solve(Y, D0, 0.1, **kw)
    assert it == it2 == 3
    assert torch.equal(D, D2) and torch.equal(x, x2)
    assert bool(torch.isfinite(D).all()) and bool(torch.isfinite(x).all())
    norms = torch.sqrt(torch.sum(D * D, dim=1))
    assert float(norms.max()) <= 1.0 + 1e-4            # normalize.l2: |D_k| <= 1
    dens = float((x != 0).float().mean())
    assert 0.005 < dens < 0.3, dens
    x_start = torch.ones_like(x)
    assert _dl_objective(Y, x, D, 0.1) < _dl_objective(Y, x_start, D0, 0.1)


def test_c5_complex_dictionary_step_properties():
    """BASELINE configs[4] at one GPU's minibatch shape (8192 x 8192 complex64, k = 512): one epoch over
    two minibatches; finite, sparse, deterministic.  SECONDARY to
    test_dictionary_step_oracle_parity_ista_at_timed_shape (the NumPy oracle at this shape): the
    complex64 result also agrees with the same run in complex128 (fp64 MFMA core) to single precision."""
    import torch
    import decomp_amd
    rows, F_, K_ = 2 * 8192, 8192, 512
    Y, D0 = _dl_data(rows, F_, K_, cplx=True, seed=3)
    kw = dict(tol=0.0, minibatch=8192, maxiter=2, lasso_method='ista', lasso_iter=10, lasso_tol=1e-5,
              random_seed=0)
    it, D, x = decomp_amd.dictionary_learning.solve(Y, D0, 0.1, **kw)
    it2, D2, x2 = decomp_amd.dictionary_learning.solve(Y, D0, 0.1, **kw)
    assert it == it2 == 2 and torch.equal(D, D2) and torch.equal(x, x2)
    assert bool(torch.isfinite(torch.view_as_real(D)).all()) and bool(torch.isfinite(torch.view_as_real(x)).all())
    dens = float((x != 0).float().mean())
    assert 0.005 < dens < 0.3, dens
    itd, Dd, xd = decomp_amd.dictionary_learning.solve(Y.to(torch.complex128), D0.to(torch.complex128), 0.1, **kw)
    scale = float(torch.abs(Dd).max())
    assert float(torch.abs(D.to(torch.complex128) - Dd).max()) < 2e-3 * scale
    # codes: same support up to threshold-crossers, same values elsewhere
    both = (x != 0) & (xd != 0)
    assert float(both.float().sum()) > 0.98 * float((xd != 0).float().sum())
    assert float(torch.abs(x.to(torch.complex128) - xd)[both].max()) < 2e-2 * float(torch.abs(xd).max())


def test_c2_fp32_residual_trace_matches_fp64_to_1e5():
    """north_star: "per-iteration residual matching NumPy to 1e-5 rel" at the FULL configs[1] shape,
    secondary evidence next to test_c2_oracle_parity_full_shape, over more iterations: the float32 path (fp32 MFMA, Gram formulation) against
    the same data iterated in float64 (fp64 MFMA core, itself checked against the oracle at small sizes),
    ||Y - x D||_F after each of the first 6 iterations."""
    import ctypes
    import torch
    from decomp_amd import _arrays, _hip
    Y, D0 = _data(N, seed=11)
    n_it = 6

    def trace(Yt, Dt, sfx, ctype):
        D = Dt.clone()
        _arrays.l2_normalize_(D, strict=True)
        x = torch.ones((N, K), device='cuda', dtype=Yt.dtype)
        lib, h = _arrays.lib_handle(Yt)
        tr = (ctype * (n_it + 1))()
        it = ctypes.c_int(0)
        fn = getattr(lib, 'dcp_nmf_mu_' + sfx)
        _hip.check(h, fn(h, _arrays.ptr(Yt), None, _arrays.ptr(x), _arrays.ptr(D), N, F, K, _hip.LIK_L2,
                         ctype(0.0), n_it + 1, ctypes.byref(it), None, tr), 'nmf_mu_' + sfx)
        assert it.value == n_it + 1
        return np.array([tr[i] for i in range(n_it)], dtype=np.float64)
    r32 = trace(Y, D0, 'f32', ctypes.c_float)
    r64 = trace(Y.double(), D0.double(), 'f64', ctypes.c_double)
    rel = np.abs(r32 - r64) / r64
    assert np.all(np.diff(r64) <= 0)            # MU never increases the loss
    assert rel.max() <= 1e-5, rel


@pytest.mark.parametrize('lik', ['l2', 'kl'])
def test_c4_shard_masked_fp32_trace_matches_fp64(lik):
    """The same check for one 16384-row shard of the masked configs[3] (20 % missing), l2 and kl:
    masked residual ||(Y - x D) o M||_F per iteration, float32 vs float64."""
    import ctypes
    import torch
    from decomp_amd import _arrays, _hip
    rows = 16384
    Y, D0 = _data(rows, seed=13)
    g = torch.Generator(device='cuda')
    g.manual_seed(17)
    mask = (torch.rand((rows, F), generator=g, device='cuda') >= 0.2).float()
    n_it = 5
    code = _hip.LIK_L2 if lik == 'l2' else _hip.LIK_KL

    def trace(Yt, Mt, Dt, sfx, ctype):
        D = Dt.clone()
        _arrays.l2_normalize_(D, strict=True)
        x = torch.ones((rows, K), device='cuda', dtype=Yt.dtype)
        lib, h = _arrays.lib_handle(Yt)
        tr = (ctype * (n_it + 1))()
        it = ctypes.c_int(0)
        fn = getattr(lib, 'dcp_nmf_mu_' + sfx)
        _hip.check(h, fn(h, _arrays.ptr(Yt), _arrays.ptr(Mt), _arrays.ptr(x), _arrays.ptr(D), rows, F, K,
                         code, ctype(0.0), n_it + 1, ctypes.byref(it), None, tr), 'nmf_mu_' + sfx)
        return np.array([tr[i] for i in range(n_it)], dtype=np.float64)
    r32 = trace(Y, mask, D0, 'f32', ctypes.c_float)
    r64 = trace(Y.double(), mask.double(), D0.double(), 'f64', ctypes.c_double)
    rel = np.abs(r32 - r64) / r64
    assert rel.max() <= 1e-5, rel


def test_more_than_2_pow_32_elements():
    """Y 1048576 x 4096 (4.3e9 elements, 17 GB), k = 64: every index computation beyond 32 bits.  Two MU
    iterations decrease the residual, and the last 4096 rows carry exactly their share of it (an index that
    wrapped would leave the upper half of x untouched or garbage)."""
    import ctypes
    import torch
    from decomp_amd import _arrays, _hip
    n, f, k = 1 << 20, 4096, 64
    g = torch.Generator(device='cuda')
    g.manual_seed(0)
    Dt = torch.rand((k, f), generator=g, device='cuda')
    xt = torch.rand((n, k), generator=g, device='cuda')
    Y = torch.empty((n, f), device='cuda')
    for r0 in range(0, n, 1 << 16):
        Y[r0:r0 + (1 << 16)] = xt[r0:r0 + (1 << 16)] @ Dt
    del xt
    Y += 0.05 * torch.rand((n, f), generator=g, device='cuda')
    D = Dt + 0.3 * torch.rand((k, f), generator=g, device='cuda')
    _arrays.l2_normalize_(D, strict=True)
    x = torch.ones((n, k), device='cuda')
    lib, h = _arrays.lib_handle(D)
    it = ctypes.c_int(0)

    def resid():
        out = ctypes.c_double(0)
        _hip.check(h, lib.dcp_nmf_residual_f32(h, _arrays.ptr(Y), None, _arrays.ptr(x), _arrays.ptr(D), n, f, k,
                                               ctypes.byref(out)), 'resid')
        return out.value
    r_prev = resid()
    for _ in range(2):
        _hip.check(h, lib.dcp_nmf_mu_f32(h, _arrays.ptr(Y), None, _arrays.ptr(x), _arrays.ptr(D), n, f, k,
                                         _hip.LIK_L2, ctypes.c_float(0.0), 2, ctypes.byref(it), None, None), 'nmf')
        r = resid()
        assert r < r_prev and np.isfinite(r)
        r_prev = r
    assert bool(torch.isfinite(x).all())
    tail = Y[-4096:] - x[-4096:] @ D
    r_tail = float((tail.double() ** 2).sum().sqrt())
    assert abs(r_tail - r_prev / 16.0) < 0.02 * r_tail, (r_tail, r_prev / 16.0)     # 4096 of 2^20 rows: 1/256 of the squares
    assert abs(float(x[-4096:].mean()) - float(x[:4096].mean())) < 0.01 * float(x[:4096].mean())


def test_lasso_rows_beyond_2_pow_31_elements():
    """lasso.solve on y 600000 x 4096 (2.46e9 elements): the rows of a LASSO problem are independent, so the
    last 2048 rows solved on their own must give the same codes as inside the big solve."""
    import torch
    from decomp_amd import lasso
    n, f, k = 600000, 4096, 64
    g = torch.Generator(device='cuda')
    g.manual_seed(5)
    A = torch.randn((k, f), generator=g, device='cuda')
    y = torch.empty((n, f), device='cuda')
    for r0 in range(0, n, 50000):
        xt = torch.randn((50000, k), generator=g, device='cuda') * (torch.rand((50000, k), generator=g, device='cuda') < 0.1)
        y[r0:r0 + 50000] = xt @ A
    y += 0.1 * torch.randn((n, f), generator=g, device='cuda')
    it, x = lasso.solve(y, A, 0.1, tol=1e-12, method='ista', maxiter=6)
    it2, x_tail = lasso.solve(y[-2048:].clone(), A, 0.1, tol=1e-12, method='ista', maxiter=6)
    assert it == it2
    assert bool(torch.isfinite(x).all())
    d = float((x[-2048:] - x_tail).abs().max())
    assert d <= 1e-6 * max(1.0, float(x_tail.abs().max())), d
    assert int((x[-2048:] != 0).sum()) > 0
