"""GPU: BASELINE.json's full sizes through size-independent properties (the oracle cannot
run these shapes in seconds): Y 65536 x 4096, k = 256, float32 (configs[1]) and one
16384-row shard of the masked configs[3].  Data are synthesised on the GPU with torch
(test plumbing only)."""
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
